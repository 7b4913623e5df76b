#!/usr/bin/env python3
"""Which registers go wrong?  (round 5; VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_vfdbg.so, built with -DVFX_VF_DEBUG)
k_voc_final keeps every sum twice in separate registers and logs the lanes whose copies differ before (code 1) or after
(code 2) the group reduction.  Victim: the vocoder without ResStacks on stream A; culprit: the other engine's mel ResUNet on B."""
import ctypes
import os
import struct
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
    ev = Engine(dev, config={"precision": 2, "tuning": 3, "voc_depth": [0] * 8})
    ev.load_state_dict(MODEL_VOCODER, vsd)
    eu = Engine(dev, config={"precision": 2})
    eu.load_state_dict(MODEL_UNET_MEL, usd)
    lib = ev.lib
    wavs = [base[:, :100000 + 20000 * k].contiguous() for k in range(6)]
    mels = [eu.stft(w)["mel"] for w in wavs]
    ref = [ev.vocoder(m) for m in mels]
    torch.cuda.synchronize()

    def log():
        buf = (ctypes.c_uint32 * (1 + 8 * 4096))()
        rc = lib.vfx_debug_read_vf(buf, 1 + 8 * 4096)
        a = np.frombuffer(buf, dtype=np.uint32)
        return rc, int(a[0]), a[1:].reshape(4096, 8)

    rc, n, _ = log()
    print("sequential run: rc %d, %d log entries" % (rc, n), flush=True)
    lib.vfx_debug_reset_vf()
    nbad = 0
    for rep in range(4):
        outs = []
        torch.cuda.synchronize()
        for i, m in enumerate(mels):
            with torch.cuda.stream(sb):
                eu.resunet_mel(m)
            with torch.cuda.stream(sa):
                outs.append(ev.vocoder(m))
        torch.cuda.synchronize()
        nbad += sum(int((a != b).sum()) for a, b in zip(outs, ref))
    rc, n, rows = log()
    print("two streams: %d wrong output samples; %d log entries" % (nbad, n), flush=True)
    f = lambda u: struct.unpack("f", struct.pack("I", int(u)))[0]
    codes = {}
    for r in rows[:min(n, 4096)]:
        codes[int(r[0])] = codes.get(int(r[0]), 0) + 1
    print("entries by code:", codes, flush=True)
    for r in rows[:min(n, 60)]:
        print("code %d block (%d, %d) thread %d (lane %d) o/g %d: copy A %.9g (0x%08x) copy B %.9g (0x%08x) diff %.3g" %
              (r[0], r[1], r[2], r[3], r[3] % 64, r[4], f(r[5]), r[5], f(r[6]), r[6], f(r[5]) - f(r[6])), flush=True)


if __name__ == "__main__":
    main()
