#!/usr/bin/env python3
"""A test set of clips of UNEQUAL length (the reference iterates files of any length, one handler call each:
evaluation_proc/eval.py:119-134) through `VoiceFixer.restore_list`, timed three ways on the same clips, resident in HBM:

  varlen      the shipped path: padded batches of up to 37 clips of ANY lengths through `vfx_restore_gsr_varlen` (round 6: the
              ResUNet once per padded frame count inside the call, ONE vocoder pass; round 5: one call per padded frame count);
  per_clip    one `restore` call per clip (what rounds 1-4 did for a real test set: equal lengths are rare);
  equal       the same amount of audio as ONE batch of equal-length clips (`restore`; the configs[1] shape when the total
              is 160 s) -- the ceiling a padded batch is compared with.

    python scripts/bench_varlen.py [--clips=32] [--lo=2] [--hi=8] [--precision=2] [--reps=3] > gpurun_out/varlen.json
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicefixer_main_amd import models, synth  # noqa: E402
from voicefixer_main_amd.engine import Engine  # noqa: E402


def opt(name, default, cast=float):
    v = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--%s=" % name)]
    return cast(v[0]) if v else default


def main():
    n, lo, hi = opt("clips", 32, int), opt("lo", 2.0), opt("hi", 8.0)
    precision, reps = opt("precision", 2, int), opt("reps", 3, int)
    dev = torch.device("cuda:0")
    eng = Engine(dev, config={"precision": precision})
    m = models.VoiceFixer(None, channels=2, type_target="vocals", engine=eng)
    sd = {"generator.analysis_module." + k: v for k, v in synth.make_resunet_state_dict(0).items()}
    sd.update({"vocoder." + k: v for k, v in synth.make_vocoder_state_dict(1).items()})
    m.load_state_dict(sd)
    rng = np.random.default_rng(2025)
    lens = [int(v) for v in rng.uniform(lo * 44100, hi * 44100, size=n)]      # VCTK-shaped: 2 .. 8 s, no two alike
    base = synth.make_clips(n, hi + 0.1, seed=77)[:, 0]
    clips = [torch.from_numpy(base[i, :L].copy()).to(dev) for i, L in enumerate(lens)]
    total = sum(lens) / 44100.0
    res = {"clips": n, "seconds_min_max": [round(min(lens) / 44100.0, 2), round(max(lens) / 44100.0, 2)],
           "audio_seconds": round(total, 1), "precision": precision,
           "buckets": sorted({eng.padded_frames(L) for L in lens})}

    def timed(fn):
        fn()                                   # plans, arena
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), out

    dt, got = timed(lambda: m.restore_list(clips))
    res["varlen_s"], res["varlen_audio_s_per_s"] = round(dt, 4), round(total / dt, 1)
    if "--only-varlen" in sys.argv:            # (a profiler run: nothing but the shipped path in the trace)
        print(json.dumps(res))
        return
    dt1, one = timed(lambda: [m.restore(c[None])[0] for c in clips])
    res["per_clip_s"], res["per_clip_audio_s_per_s"] = round(dt1, 4), round(total / dt1, 1)
    res["varlen_equals_per_clip"] = bool(all(torch.equal(a, b) for a, b in zip(got, one)))
    if not res["varlen_equals_per_clip"]:      # diagnostics: which clips, where, how much; is either path repeatable?
        again_v, again_1 = m.restore_list(clips), [m.restore(c[None])[0] for c in clips]
        res["varlen_repeatable"] = bool(all(torch.equal(a, b) for a, b in zip(got, again_v)))
        res["per_clip_repeatable"] = bool(all(torch.equal(a, b) for a, b in zip(one, again_1)))
        bad = []
        for i, (a, b) in enumerate(zip(got, one)):
            if not torch.equal(a, b):
                d = (a - b).abs()
                nz = torch.nonzero(d > 0)
                bad.append({"clip": i, "samples": lens[i], "frames": lens[i] // 441 + 1, "padded": eng.padded_frames(lens[i]),
                            "max_abs_diff": float(d.max()), "first_diff_at": int(nz[0]), "last_diff_at": int(nz[-1]),
                            "n_diff": int(nz.shape[0]), "peak": float(b.abs().max())})
        res["mismatches"] = bad[:12]
        res["n_mismatch"] = len(bad)
    L_eq = int(round(total / 16 * 44100))
    eq = torch.from_numpy(synth.make_clips(16, L_eq / 44100.0, seed=78)[:, 0, :L_eq].copy()).to(dev)
    dt2, _ = timed(lambda: m.restore(eq))
    res["equal_batch_s"], res["equal_batch_audio_s_per_s"] = round(dt2, 4), round(16 * L_eq / 44100.0 / dt2, 1)
    res["varlen_vs_equal"] = round(res["varlen_audio_s_per_s"] / res["equal_batch_audio_s_per_s"], 3)
    res["flags"] = int(eng.take_flags())
    print(json.dumps(res))


if __name__ == "__main__":
    main()
