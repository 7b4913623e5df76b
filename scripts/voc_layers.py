#!/usr/bin/env python3
"""A/B harness of the vocoder's fused ResStack layers: runs the vocoder (16 x 10 s, precision 2 unless --precision=1) with the
library named by VFX_LIB_PATH / the VFX_* switches of the environment, HIP events around every GEMM-shaped launch
(VFX_PROFILE_DUMP), and prints one line per (kernel, dilation): median / min ms over the repeats, plus the per-stack sums.

    [VFX_LIB_PATH=...] python scripts/voc_layers.py TAG [--reps=5] [--precision=2] [--tuning=MASK] [--json=out.jsonl]
"""
import collections
import csv
import json
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicefixer_main_amd import synth  # noqa: E402
from voicefixer_main_amd.engine import Engine, MODEL_VOCODER  # noqa: E402


def opt(name, default):
    v = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--%s=" % name)]
    return v[0] if v else default


def main():
    tag = ([a for a in sys.argv[1:] if not a.startswith("--")] or ["run"])[0]
    reps, precision = int(opt("reps", "5")), int(opt("precision", "2"))
    eng = Engine("cuda:0", config={"precision": precision, "tuning": int(opt("tuning", "0"), 0)})
    eng.load_state_dict(MODEL_VOCODER, synth.make_vocoder_state_dict(1))
    g = torch.Generator(device="cuda").manual_seed(1)
    mel = torch.rand(16, 1001, 128, device="cuda", generator=g) * 0.1
    out = eng.vocoder(mel)
    torch.cuda.synchronize()
    dump = tempfile.NamedTemporaryFile(suffix=".csv", delete=False).name
    os.environ["VFX_PROFILE_DUMP"] = dump
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eng.profile_begin()
    ev0.record()
    for _ in range(reps):
        out = eng.vocoder(mel)
    ev1.record()
    n, ms, fl = eng.profile_end()
    torch.cuda.synchronize()
    rows = list(csv.DictReader(open(dump)))
    os.unlink(dump)
    per = collections.OrderedDict()
    for r in rows:
        k = (r["kernel"].replace(";", ","), int(r["Wi"]) if r["kernel"].startswith("k_resblock") else 0, int(r["M"]), int(r["Cout"]))
        per.setdefault(k, []).append((float(r["ms"]), float(r["tflops"])))
    stacks = collections.OrderedDict()
    import hashlib
    sha = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]      # bit-identity of two builds on the same input
    res = {"tag": tag, "precision": precision, "reps": reps, "checksum": float(out.double().abs().sum().item()), "sha256_16": sha,
           "finite": bool(torch.isfinite(out).all().item()), "flags": eng.take_flags(), "layers": []}
    print("== %s (precision %d, %d reps) checksum %.9e sha256 %s finite %s" % (tag, precision, reps, res["checksum"], sha, res["finite"]))
    for (k, d, M, C), v in per.items():
        t = sorted(x[0] for x in v)
        med, mn = t[len(t) // 2], t[0]
        calls = len(v) // reps
        if k.startswith("k_resblock"):
            print("  %-34s d=%-5d x%d  median %.4f ms  min %.4f  (%.0f TF)" % (k, d, calls, med, mn, sorted(x[1] for x in v)[len(v) // 2]))
            res["layers"].append({"kernel": k, "dil": d, "median_ms": med, "min_ms": mn})
        elif "--convs" in sys.argv:      # the tap-convolution launches too (condnet, k7, upsamplers, the C = 512 stack): M, Cout
            print("  %-34s M=%-8d Cout=%-5d x%d  median %.4f ms  min %.4f  (%.0f TF)" % (k, M, C, calls, med, mn, sorted(x[1] for x in v)[len(v) // 2]))
        s = stacks.setdefault(k, [0.0, 0])
        s[0] += sum(t) / reps
        s[1] += calls
    print("  -- per kernel, ms per vocoder call:")
    for k, (t, c) in stacks.items():
        print("  %-34s x%-3d %8.3f ms" % (k, c, t))
    res["per_kernel_ms"] = {k: round(t, 4) for k, (t, c) in stacks.items()}
    res["conv_ms_per_call"] = round(ms / reps, 3)
    res["wall_ms_per_call"] = round(ev0.elapsed_time(ev1) / reps, 3)
    print("  GEMM-shaped launches: %.3f ms per call (events), %.3f ms per call (wall incl. event overhead)" % (
        res["conv_ms_per_call"], res["wall_ms_per_call"]))
    j = opt("json", "")
    if j:
        with open(j, "a") as f:
            f.write(json.dumps(res) + "\n")


if __name__ == "__main__":
    main()
