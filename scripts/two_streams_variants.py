#!/usr/bin/env python3
"""Victim and culprit variants (round 5; VFX_LIB_PATH=.../libvfx_vfdbg.so).  Victims: k_voc_final on the fp16 trunk (precision 2) and
on the fp32 trunk (precision 1: no 16-bit instructions in it).  Culprits on the other stream: the STFT front end (FFT kernels: LDS,
no MFMA, no LDS-DMA), the mel ResUNet in fp32 (32x32x2 fp32 MFMA), in split-bf16 (32x32x16 bf16 MFMA), the vocoder without ResStacks
(32x32x16 fp16 MFMA).  The count is k_voc_final's own log: lanes whose two copies of a sum differ."""
import ctypes
import os
import sys

import numpy as np
import torch

os.environ.setdefault("VFX_NO_STREAM_TURNS", "1")      # the measurement needs the launches of the two streams to overlap

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicefixer_main_amd import synth  # noqa: E402
from voicefixer_main_amd.engine import Engine, MODEL_UNET_MEL, MODEL_VOCODER  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    usd, vsd = synth.make_resunet_state_dict(0), synth.make_vocoder_state_dict(1)
    base = torch.from_numpy(synth.make_clips(13, 7.0, seed=5)[:, 0]).to(dev)
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    victims = {}
    for prec in (2, 1):
        e = Engine(dev, config={"precision": prec, "tuning": 3, "voc_depth": [0] * 8})
        e.load_state_dict(MODEL_VOCODER, vsd)
        victims["k_voc_final precision %d" % prec] = e
    culprit_engines = {}
    for prec in (0, 1):
        e = Engine(dev, config={"precision": prec})
        e.load_state_dict(MODEL_UNET_MEL, usd)
        culprit_engines[prec] = e
    ec = Engine(dev, config={"precision": 2, "tuning": 3, "voc_depth": [0] * 8})
    ec.load_state_dict(MODEL_VOCODER, vsd)
    lib = ec.lib
    wavs = [base[:, :100000 + 20000 * k].contiguous() for k in range(6)]
    mels = [ec.stft(w)["mel"] for w in wavs]
    culprits = {
        "nothing": lambda i: None,
        "stft": lambda i: [culprit_engines[1].stft(wavs[i]) for _ in range(3)],
        "mel ResUNet fp32 MFMA": lambda i: culprit_engines[0].resunet_mel(mels[i]),
        "mel ResUNet split-bf16 MFMA": lambda i: culprit_engines[1].resunet_mel(mels[i]),
        "vocoder without ResStacks, fp16 MFMA": lambda i: [ec.vocoder(mels[i]) for _ in range(2)],
    }

    def count():
        buf = (ctypes.c_uint32 * 1)()
        lib.vfx_debug_read_vf(buf, 1)
        return int(buf[0])

    for vn, ev in victims.items():
        ref = [ev.vocoder(m) for m in mels]
        torch.cuda.synchronize()
        for cn, cul in culprits.items():
            cul(0)
            torch.cuda.synchronize()
            lib.vfx_debug_reset_vf()
            wrong = 0
            for _ in range(4):
                outs = []
                torch.cuda.synchronize()
                for i, m in enumerate(mels):
                    with torch.cuda.stream(sb):
                        cul(i)
                    with torch.cuda.stream(sa):
                        outs.append(ev.vocoder(m))
                torch.cuda.synchronize()
                wrong += sum(int((a != b).sum()) for a, b in zip(outs, ref))
            n = count()
            if cn.startswith("vocoder"):
                n_note = " (the culprit's own k_voc_final logs here too)"
            else:
                n_note = ""
            print("%s beside %s: %d wrong output samples, %d log entries%s" % (vn, cn, wrong, n, n_note), flush=True)


if __name__ == "__main__":
    main()
