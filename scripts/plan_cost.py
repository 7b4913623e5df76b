#!/usr/bin/env python3
"""What a plan-cache MISS costs: the same padded batch (13 clips, ~5 s) through vfx_restore_gsr_varlen with a row length that
repeats (hit) and one that is new on every call (miss: the plan is built -- geometry, stage tables, parameter upload -- before the
call can be enqueued), wall time per call with the queue kept full (one synchronize at the end of 12 calls)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicefixer_main_amd import synth  # noqa: E402
from voicefixer_main_amd.engine import Engine, MODEL_UNET_MEL, MODEL_VOCODER  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    eng = Engine(dev, config={"precision": 2})
    eng.load_state_dict(MODEL_UNET_MEL, synth.make_resunet_state_dict(0))
    eng.load_state_dict(MODEL_VOCODER, synth.make_vocoder_state_dict(1))
    B, n = 13, 12
    rng = np.random.default_rng(1)
    lens = sorted([int(v) for v in rng.uniform(203000, 224900, size=B)], reverse=True)    # frames 461 .. 511: one bucket (512)
    base = torch.from_numpy(synth.make_clips(B, 5.2, seed=5)[:, 0]).to(dev)
    res = {}
    for tag, widths in (("hit", [225700] * n), ("miss", [225000 + 7 * i for i in range(n)])):
        xs = [base[:, :w].contiguous() for w in widths]
        eng.restore_gsr_varlen(xs[0] if tag == "hit" else base[:, :224990].contiguous(), lens)     # arena, first plan
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for x in xs:
            eng.restore_gsr_varlen(x, lens)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res[tag] = {"ms_per_call_wall": round((t2 - t0) / n * 1e3, 3), "ms_per_call_host_enqueue": round((t1 - t0) / n * 1e3, 3)}
    res["audio_seconds_per_call"] = round(sum(lens) / 44100.0, 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
