#!/usr/bin/env python3
"""The drop-in boundary's own cost: `handlers.handler` (eval_gsr_voicefixer.py:37-77 line by line -- B = 1, 60-s
segments, the per-stage calls of the reference surface with their host syncs: to_log's assert, the peak compare,
wav file I/O) timed on a 150-s file, next to the fused `VoiceFixer.restore` on the same segments.

    python scripts/bench_handler.py [--precision 1|2] > gpurun_out/handler.json
"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicefixer_main_amd import handlers, models, synth  # noqa: E402
from voicefixer_main_amd.engine import Engine  # noqa: E402


def main():
    precision = int(([a.split("=")[1] for a in sys.argv[1:] if a.startswith("--precision=")] or ["1"])[0])
    dev = torch.device("cuda:0")
    eng = Engine(dev, config={"precision": precision})
    m = models.VoiceFixer(None, channels=2, type_target="vocals", engine=eng)
    sd = {"generator.analysis_module." + k: v for k, v in synth.make_resunet_state_dict(0).items()}
    sd.update({"vocoder." + k: v for k, v in synth.make_vocoder_state_dict(1).items()})
    m.load_state_dict(sd)
    handlers._state["model"] = m
    seconds = 150.0
    wav = synth.make_clips(1, seconds, seed=3)[0, 0]
    tmp = tempfile.mkdtemp()
    src, dst = os.path.join(tmp, "in.wav"), os.path.join(tmp, "out.wav")
    handlers.save_wave(wav, src)
    res = {"file_seconds": seconds, "segments": 3, "precision": precision}
    for tag, target in (("no_target", None), ("with_target_metrics", src)):
        handlers.handler(src, dst, target, ckpt=None, device=dev, needrefresh=False, meta={"unify_energy": False})   # warm-up: plans, arena
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):      # five calls, the MEDIAN is reported (round 5: single calls spread 43 .. 108 ms on one code state --
            t0 = time.perf_counter()   # the host's file I/O and allocator, not the GPU's 39 ms)
            handlers.handler(src, dst, target, ckpt=None, device=dev, needrefresh=False, meta={"unify_energy": False})
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        dt = float(np.median(ts))
        res["handler_%s_s" % tag] = round(dt, 4)
        res["handler_%s_calls_s" % tag] = [round(t, 4) for t in ts]
        res["handler_%s_audio_s_per_s" % tag] = round(seconds / dt, 1)
    if "--profile" in sys.argv:      # where the host's part of the handler goes (cumulative seconds per function, one call)
        import cProfile
        import io
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        handlers.handler(src, dst, None, ckpt=None, device=dev, needrefresh=False, meta={"unify_energy": False})
        torch.cuda.synchronize()
        pr.disable()
        buf = io.StringIO()
        pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(18)
        sys.stderr.write(buf.getvalue())
    # the same three segments through the fused entry point, resident in HBM (no file I/O)
    x = torch.from_numpy(wav).to(dev)
    segs = [x[i * 2646000:(i + 1) * 2646000][None] for i in range(3)]
    for s in segs:
        m.restore(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in segs:
        m.restore(s)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res["fused_restore_s"] = round(dt, 4)
    res["fused_restore_audio_s_per_s"] = round(seconds / dt, 1)
    # wav I/O share of the handler
    t0 = time.perf_counter()
    y = handlers.load_wav(src)
    handlers.save_wave(y, dst)
    res["wav_io_s"] = round(time.perf_counter() - t0, 4)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
