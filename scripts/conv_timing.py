#!/usr/bin/env python3
"""Where a block of k_conv spends its life in the deep ResUNet levels: per-wave stamps of a -DVFX_TIMING build of conv.hip
(scripts/build_variant.sh convtiming conv.hip -DVFX_TIMING), one 3 x 3 convolution per shape through vfx_op_conv.

    VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_convtiming.so python scripts/conv_timing.py [--json=out.json]

Stamps per wave (conv.hip, KCONV_TS): 0 entry, 1 setup done (parameters read, output table, per-lane geometry), 2 first patch + weights
landed, 3 first transform done, 4 tap loop done, 5 epilogue done (shader clock); block entry / exit on the chip-wide 100 MHz clock.
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

TAPS = ["barrier + fetch + patch request + weight wait", "fragment reads + MFMAs", "stage end (wait, transform) + cursors"]
PH = ["setup (params, tables)", "first patch + weights", "first transform", "tap loop (all stages)", "epilogue"]


def main():
    assert "timing" in os.environ.get("VFX_LIB_PATH", ""), "run with VFX_LIB_PATH=.../abl/libvfx_convtiming.so"
    buf = torch.zeros(48 * 196608, dtype=torch.int64, device="cuda")      # [block][wave][12] u64
    os.environ["VFX_CONV_TIMING_PTR"] = hex(buf.data_ptr())
    from voicefixer_main_amd.engine import Engine
    eng = Engine("cuda:0", config={"precision": 1})
    g = torch.Generator().manual_seed(0)
    out = {}
    # (B, H, W, Cin, Cout): level 6 and the bottleneck of a 16 x 10 s batch, level 5, level 6 of the 1-s chunk, level 4
    for B, H, W, Cin, Cout in ((16, 32, 3, 384, 384), (16, 16, 1, 384, 384), (16, 64, 7, 384, 384), (1, 4, 3, 384, 384), (16, 128, 15, 256, 256)):
        x = torch.randn((B, H, W, Cin), generator=g).cuda()
        w = (torch.randn((Cout, Cin, 3, 3), generator=g) * 0.02).numpy()
        sc, sh = np.ones(Cin, np.float32), np.zeros(Cin, np.float32)
        for _ in range(2):
            eng.op_conv(x, w, scale=sc, shift=sh, act=1, slope=0.01)
        buf.zero_()
        torch.cuda.synchronize()
        eng.op_conv(x, w, scale=sc, shift=sh, act=1, slope=0.01)
        torch.cuda.synchronize()
        ts = buf.cpu().numpy().astype(np.uint64).reshape(-1, 4, 12)
        used = ts[:, 0, 7] != 0
        ts = ts[used].astype(np.float64)
        n = ts.shape[0]
        cyc = ts[:, :, 5] - ts[:, :, 0]
        rt = (ts[:, :, 7] - ts[:, :, 6]) * 10.0                       # ns
        ghz = float((cyc / np.maximum(rt, 1.0)).mean())
        d = np.diff(ts[:, :, :6], axis=2) / ghz / 1e3                # us
        t0 = (ts[:, :, 6].min(axis=1) - ts[:, :, 6].min()) / 100.0    # block starts, us after the first
        t1 = (ts[:, :, 7].max(axis=1) - ts[:, :, 6].min()) / 100.0
        res = {"blocks": int(n), "shader_ghz": round(ghz, 3), "kernel_span_us": float(t1.max()), "block_life_us_mean": float((t1 - t0).mean()),
               "block_life_us_max": float((t1 - t0).max()), "last_block_start_us": float(t0.max()),
               "phases_us": {PH[i]: float(d[:, :, i].mean()) for i in range(5)}}
        out["B%d_%dx%d_C%d_%d" % (B, H, W, Cin, Cout)] = res
        print("== B = %d, %d x %d pixels, %d -> %d channels: %d blocks, shader clock %.2f GHz; kernel span %.1f us, block life mean %.1f / max %.1f us, "
              "last block starts at %.1f us" % (B, H, W, Cin, Cout, n, ghz, t1.max(), (t1 - t0).mean(), (t1 - t0).max(), t0.max()))
        q = np.percentile(t0, [50, 90, 99])
        print("   block starts (us after the first): median %.1f, 90 %% %.1f, 99 %% %.1f" % tuple(q))
        for i in range(5):
            print("   %-26s mean %6.2f us   slowest wave of a block %6.2f us" % (PH[i], d[:, :, i].mean(), d[:, :, i].max(axis=1).mean()))
        acc = ts[:, :, 8:11] / ghz / 1e3
        res["tap_loop_split_us"] = {k: float(acc[:, :, i].mean()) for i, k in enumerate(TAPS)}
        print("   inside the tap loop:  " + ";  ".join("%s %.2f us" % (k, acc[:, :, i].mean()) for i, k in enumerate(TAPS)))
    j = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--json=")]
    if j:
        json.dump(out, open(j[0], "w"), indent=1)


if __name__ == "__main__":
    main()
