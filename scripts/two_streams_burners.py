#!/usr/bin/env python3
"""What kind of work on the other stream disturbs k_voc_final?  (round 5; VFX_LIB_PATH=.../libvfx_vfdbg.so)  Synthetic co-runners
(`vfx_debug_burn`, csrc/small_ops.hip): 16-bit MFMAs on registers only, fp32 MFMAs on registers only, VALU FMAs only, 16-byte
LDS-DMA reads only -- each sized to run about as long as the victim's batch, on 2048 blocks (every CU busy) or on 64 blocks
(a quarter of the CUs).  The count is k_voc_final's own log of lanes whose two copies of a sum differ."""
import ctypes
import os
import sys
import time

import torch

os.environ.setdefault("VFX_NO_STREAM_TURNS", "1")

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicefixer_main_amd import synth  # noqa: E402
from voicefixer_main_amd.engine import Engine, MODEL_VOCODER  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    vsd = synth.make_vocoder_state_dict(1)
    base = torch.from_numpy(synth.make_clips(13, 7.0, seed=5)[:, 0]).to(dev)
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    ev = Engine(dev, config={"precision": 2, "tuning": 3, "voc_depth": [0] * 8})
    ev.load_state_dict(MODEL_VOCODER, vsd)
    lib = ev.lib
    lib.vfx_debug_burn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p]
    wavs = [base[:, :100000 + 20000 * k].contiguous() for k in range(6)]
    mels = [ev.stft(w)["mel"] for w in wavs]
    ref = [ev.vocoder(m) for m in mels]
    src = torch.randn((64 << 20) // 4, device=dev) * 0.01       # (kinds 7 and 8 write into it)
    out = torch.zeros(256, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for m in mels:
        ev.vocoder(m)
    torch.cuda.synchronize()
    victim_ms = (time.perf_counter() - t0) / len(mels) * 1e3
    print("victim: %.2f ms per batch alone" % victim_ms, flush=True)

    def burn(kind, blocks, iters):
        rc = lib.vfx_debug_burn(kind, blocks, iters, src.data_ptr(), src.numel() * 4, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc

    def count():
        buf = (ctypes.c_uint32 * 1)()
        lib.vfx_debug_read_vf(buf, 1)
        return int(buf[0])

    names = {0: "16-bit MFMA, registers only", 1: "fp32 MFMA, registers only", 2: "VALU FMA only", 3: "16-byte LDS-DMA reads only",
             4: "sleeping waves (wave slots only)", 5: "sleeping waves with 52 KB of LDS per block",
             6: "10 000 different VALU instructions per loop (~100 KB of code)", 7: "16-byte streaming stores only",
             8: "a convolution in miniature (LDS-DMA, barrier, 16-bit MFMAs from LDS, stores)"}
    kinds = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4, 5, 6, 7, 8]
    for blocks in (8192, 2048, 64):
        for kind in kinds:
            iters = 2000 if kind != 6 else 4    # calibrate to ~ the victim's time per batch
            for _ in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                burn(kind, blocks, iters)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) * 1e3
                if 0.6 * victim_ms < ms < 1.6 * victim_ms:
                    break
                iters = max(50 if kind != 6 else 1, int(iters * victim_ms / max(ms, 0.02)))
            lib.vfx_debug_reset_vf()
            wrong = 0
            for _ in range(4):
                outs = []
                torch.cuda.synchronize()
                for i, m in enumerate(mels):
                    with torch.cuda.stream(sb):
                        burn(kind, blocks, iters)
                    with torch.cuda.stream(sa):
                        outs.append(ev.vocoder(m))
                torch.cuda.synchronize()
                wrong += sum(int((a != b).sum()) for a, b in zip(outs, ref))
            print("beside %s, %d blocks x %d iterations (%.2f ms alone): %d wrong output samples, %d log entries" %
                  (names[kind], blocks, iters, ms, wrong, count()), flush=True)


if __name__ == "__main__":
    main()
