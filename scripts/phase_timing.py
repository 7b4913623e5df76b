#!/usr/bin/env python3
"""Per-phase cycle breakdown of the 4-wave ResStack kernels (resblock_w64.hip, resblock_r128.hip) from a timing build
(scripts/build_timing.sh -> voicefixer_main_amd/abl/libvfx_timing.so): one layer at the vocoder's shape (16 clips), lane 0 of
every wave stamps s_memtime at the phase boundaries.  Prints, per (C, dilation), the mean cycles of every phase over all waves
and tiles, the mean block lifetime and the spread of block start times.

    VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_timing.so python scripts/phase_timing.py [--json=out.json]
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PHASES = ["setup+request", "patch wait", "transform", "barrier", "conv1", "barrier", "h write", "barrier", "conv2", "barrier",
          "stage / pass 0", "epilogue rest"]


def main():
    assert "timing" in os.environ.get("VFX_LIB_PATH", ""), "run with VFX_LIB_PATH=.../abl/libvfx_timing.so"
    buf = torch.zeros(64 * 1024 * 1024 // 8, dtype=torch.int64, device="cuda")      # [tile][wave][16] u64
    os.environ["VFX_TIMING_PTR"] = hex(buf.data_ptr())
    from voicefixer_main_amd.engine import Engine
    eng = Engine("cuda:0", config={"precision": 2})
    out = {}
    g = torch.Generator().manual_seed(0)
    for C, T in ((256, 49294), (128, 147882)):
        B = 16
        x = torch.randn((B, T, C), generator=g).cuda()
        w1, w2 = (torch.randn((C, C, 3), generator=g) * 0.05).numpy(), (torch.randn((C, C, 3), generator=g) * 0.05).numpy()
        b1, b2 = (torch.randn((C,), generator=g) * 0.1).numpy(), (torch.randn((C,), generator=g) * 0.1).numpy()
        for d in (1, 243):
            eng.op_resblock(x, w1, b1, w2, b2, d, 0.01, True)          # warm-up (and the only plan: the entry point plans per call)
            buf.zero_()
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            eng.op_resblock(x, w1, b1, w2, b2, d, 0.01, True)
            ev1.record()
            torch.cuda.synchronize()
            ts = buf.cpu().numpy().astype(np.uint64).reshape(-1, 4, 16)
            used = ts[:, 0, 12] != 0
            ts = ts[used].astype(np.float64)
            n = ts.shape[0]
            dt = np.diff(ts[:, :, :13], axis=2)                                       # (tiles, waves, 12)
            life = ts[:, :, 12] - ts[:, :, 0]
            t0 = ts[:, :, 0].min(axis=1)
            span = ts[:, :, 12].max() - t0.min()
            res = {"tiles": int(n), "mean_block_cycles": float(life.mean()), "kernel_span_cycles": float(span),
                   "phases": {("%02d %s" % (i, PHASES[i])): float(dt[:, :, i].mean()) for i in range(12)},
                   "call_ms_incl_host_prep": ev0.elapsed_time(ev1)}
            out["C%d_d%d" % (C, d)] = res
            print("== C = %d, d = %d: %d tiles, block lifetime %.0f cycles (mean), kernel span %.0f cycles, blocks in flight = %.1f" % (
                C, d, n, life.mean(), span, n * life.mean() / span / 256.0), "per CU")
            for k, v in res["phases"].items():
                print("   %-22s %9.0f cycles  %5.1f %%" % (k, v, 100.0 * v / life.mean()))
    j = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--json=")]
    if j:
        json.dump(out, open(j[0], "w"), indent=1)


if __name__ == "__main__":
    main()
