#!/usr/bin/env python3
"""Time the mel ResUNet alone at the benched shape (16 x 1001 frames), HIP events, median of the repeats -- the A/B harness of
the ResUNet-side experiments (VFX_LIB_PATH = a variant library, --tuning=MASK).

    [VFX_LIB_PATH=...] python scripts/unet_time.py TAG [--reps=10] [--tuning=0] [--json=out.jsonl]
"""
import json
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
    j = opt("json", "")
    if j:
        with open(j, "a") as f:
            f.write(json.dumps(res) + "\n")


if __name__ == "__main__":
    main()
