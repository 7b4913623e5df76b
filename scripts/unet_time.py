#!/usr/bin/env python3
"""Time the mel ResUNet alone at the benched shape (16 x 1001 frames), HIP events, median of the repeats -- the A/B harness of
the ResUNet-side experiments (VFX_LIB_PATH = a variant library, --tuning=MASK).

    [VFX_LIB_PATH=...] python scripts/unet_time.py TAG [--reps=10] [--tuning=0] [--json=out.jsonl] [--layers]

--layers: after the timing, HIP events around every GEMM-shaped launch of five more calls (VFX_PROFILE_DUMP) and one line per group of
equal launches (kernel, M, Cout, K, input channels) in plan order: calls per ResUNet pass, median ms of one, ms per pass, TFLOP/s.
"""
import collections
import csv
import json
import tempfile
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicefixer_main_amd import synth  # noqa: E402
from voicefixer_main_amd.engine import Engine, MODEL_UNET_MEL  # noqa: E402


def opt(name, default):
    v = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--%s=" % name)]
    return v[0] if v else default


def main():
    tag = ([a for a in sys.argv[1:] if not a.startswith("--")] or ["run"])[0]
    reps = int(opt("reps", "10"))
    eng = Engine("cuda:0", config={"precision": 1, "tuning": int(opt("tuning", "0"), 0)})
    eng.load_state_dict(MODEL_UNET_MEL, synth.make_resunet_state_dict(0))
    g = torch.Generator(device="cuda").manual_seed(1)
    mel = torch.rand(16, 1001, 128, device="cuda", generator=g) * 0.1 + 1e-4
    for _ in range(3):
        out = eng.resunet_mel(mel)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        out = eng.resunet_mel(mel)
        ev[i + 1].record()
    torch.cuda.synchronize()
    t = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    res = {"tag": tag, "reps": reps, "median_ms": t[len(t) // 2], "min_ms": t[0], "checksum": float(out.double().abs().sum().item()),
           "flags": eng.take_flags()}
    print("== %s: mel ResUNet 16 x 1001 frames: median %.3f ms, min %.3f  (checksum %.6e)" % (tag, res["median_ms"], res["min_ms"], res["checksum"]))
    if "--layers" in sys.argv:
        n = 5
        dump = tempfile.NamedTemporaryFile(suffix=".csv", delete=False).name
        os.environ["VFX_PROFILE_DUMP"] = dump
        eng.profile_begin()
        for _ in range(n):
            eng.resunet_mel(mel)
        launches, ms, fl = eng.profile_end()
        rows = list(csv.DictReader(open(dump)))
        os.unlink(dump)
        per = collections.OrderedDict()
        for r in rows:
            k = (r["kernel"].replace(";", ","), int(r["M"]), int(r["Cout"]), int(r["K"]), int(r["C0"]), int(r["nseg"]))
            per.setdefault(k, []).append((float(r["ms"]), float(r["tflops"])))
        print("  %-30s %9s %5s %6s %4s %4s %5s %9s %9s %7s" % ("kernel", "M", "Cout", "K", "C0", "nseg", "calls", "median ms", "ms/pass", "TF"))
        tot, layers = 0.0, []
        for (k, M, Cout, K, C0, nseg), v in per.items():
            tms = sorted(x[0] for x in v)
            med, calls, per_pass = tms[len(tms) // 2], len(v) // n, sum(tms) / n
            tot += per_pass
            print("  %-30s %9d %5d %6d %4d %4d %5d %9.4f %9.3f %7.0f" % (k, M, Cout, K, C0, nseg, calls, med, per_pass, sorted(x[1] for x in v)[len(v) // 2]))
            layers.append({"kernel": k, "M": M, "Cout": Cout, "K": K, "C0": C0, "nseg": nseg, "calls": calls, "median_ms": med, "ms_per_pass": round(per_pass, 4)})
        print("  GEMM-shaped launches: %d per pass, %.3f ms per pass (events around each)" % (launches // n, tot))
        res["layers"] = layers
    j = opt("json", "")
    if j:
        with open(j, "a") as f:
            f.write(json.dumps(res) + "\n")


if __name__ == "__main__":
    main()
