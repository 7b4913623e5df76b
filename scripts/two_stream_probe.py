#!/usr/bin/env python3
"""Probe: 16 clips as one batch on one stream vs two half batches on two streams (two handles), staggered so that one
half's vocoder (bandwidth-bound) runs beside the other half's ResUNet (MFMA-bound)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicefixer_main_amd import synth
from voicefixer_main_amd.engine import Engine, MODEL_UNET_MEL, MODEL_VOCODER

def mk():
    e = Engine("cuda:0", config={"precision": 2})
    e.load_state_dict(MODEL_UNET_MEL, synth.make_resunet_state_dict(0))
    e.load_state_dict(MODEL_VOCODER, synth.make_vocoder_state_dict(1))
    return e
wav = torch.from_numpy(synth.make_clips(16, 10.0)[:, 0]).cuda()
out = torch.empty_like(wav)
e0 = mk()
for _ in range(3): e0.restore_gsr(wav, out=out)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): e0.restore_gsr(wav, out=out)
torch.cuda.synchronize(); one = (time.perf_counter() - t0) / 10
for nsplit in (2, 4):
    engs = [e0] + [mk() for _ in range(nsplit - 1)]
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    n = 16 // nsplit
    parts = [(wav[i * n:(i + 1) * n].contiguous(), out[i * n:(i + 1) * n]) for i in range(nsplit)]
    def step():
        for e, s, (w, o) in zip(engs, streams, parts):
            with torch.cuda.stream(s):
                e.restore_gsr(w, out=o)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); multi = (time.perf_counter() - t0) / 10
    print("1 stream x16: %.2f ms   %d streams x%d: %.2f ms" % (one * 1e3, nsplit, n, multi * 1e3))
