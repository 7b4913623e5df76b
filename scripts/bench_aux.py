#!/usr/bin/env python3
"""Auxiliary measurements for the BASELINE.json configs that are not the bench line (bench.py = configs[1]):

  configs[2]  ssr_unet super-resolution, batch = 64 x 3 s             (spectrogram ResUNet + 2 STFT + ISTFT)
  configs[3]  per-GPU shard of the 1024-clip job: 128 x 10 s restore  (same path as bench.py, 8x the batch)
  configs[4]  gsr_unet streaming, 1-s chunks, hipGraph-captured step  (spectrogram ResUNet, B = 1, T = 101)

One JSON object per line.  Synthetic clips, seeded random weights; inputs resident in HBM.
    python scripts/bench_aux.py [ssr] [shard] [stream] [--precision=1|2]
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicefixer_main_amd import synth  # noqa: E402
from voicefixer_main_amd.engine import Engine, MODEL_UNET_MEL, MODEL_UNET_SPEC, MODEL_VOCODER  # noqa: E402

MAC_SPEC_PER_FRAME = 780251136


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--precision=")]
    precision = int(([a.split("=")[1] for a in sys.argv[1:] if a.startswith("--precision=")] or ["2"])[0])
    which = args or ["ssr", "shard", "stream"]
    eng = Engine("cuda:0", config={"precision": precision})   # 2 = bench.py's mode (only the vocoder differs from 1)
    eng.load_state_dict(MODEL_UNET_MEL, synth.make_resunet_state_dict(0))
    eng.load_state_dict(MODEL_VOCODER, synth.make_vocoder_state_dict(1))
    eng.load_state_dict(MODEL_UNET_SPEC, synth.make_resunet_state_dict(2))
    if "ssr" in which:
        B, sec = 64, 3.0
        wav = torch.from_numpy(synth.make_clips(B, sec, seed=7, mode="lowpass")[:, 0]).cuda()
        def step():
            sp = eng.stft(wav, want_mel=False, want_sp=True)["sp"]      # eval_ssr_unet.py:80
            return eng.resunet_spec(sp, wav)                            # STFT (phase) + trunk + ISTFT, unet_v2.py:86-148
        dt = timed(step, 3, 1)
        T = wav.shape[1] // 441 + 1
        Tpad = (T + 63) // 64 * 64
        print(json.dumps({"config": "ssr_unet SR batch=64x3s", "ms_per_step": round(dt * 1e3, 2), "audio_s_per_s": round(B * sec / dt, 1),
                          "trunk_tflops": round(2 * MAC_SPEC_PER_FRAME * Tpad * B / dt / 1e12, 1),
                          "finite": bool(torch.isfinite(step()).all())}))
        del wav
        torch.cuda.empty_cache()
    if "shard" in which:
        B, sec = 128, 10.0
        wav = torch.from_numpy(synth.make_clips(B, sec, seed=9)[:, 0]).cuda()
        out = torch.empty_like(wav)
        dt = timed(lambda: eng.restore_gsr(wav, out=out), 2, 1)
        print(json.dumps({"config": "gsr_voicefixer shard 128x10s (1/8 of the 1024-clip job)", "ms_per_step": round(dt * 1e3, 1),
                          "audio_s_per_s": round(B * sec / dt, 1), "finite": bool(torch.isfinite(out).all()),
                          "max_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}))
        del wav, out
        torch.cuda.empty_cache()
    if "stream" in which:
        chunk = torch.from_numpy(synth.make_clips(1, 1.0, seed=11)[:, 0]).cuda()
        def step():
            sp = eng.stft(chunk, want_mel=False, want_sp=True)["sp"]
            return eng.resunet_spec(sp, chunk)
        eager = timed(step, 20, 3)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            step()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=side):
                y = step()
        torch.cuda.current_stream().wait_stream(side)
        graph = timed(g.replay, 50, 5)
        print(json.dumps({"config": "gsr_unet streaming 1-s chunk, B=1", "eager_ms_per_chunk": round(eager * 1e3, 3),
                          "hipgraph_ms_per_chunk": round(graph * 1e3, 3), "real_time_factor_inv": round(1.0 / graph, 1),
                          "trunk_tflops": round(2 * MAC_SPEC_PER_FRAME * 128 / graph / 1e12, 1), "finite": bool(torch.isfinite(y).all())}))


if __name__ == "__main__":
    main()
