#!/usr/bin/env python3
"""CPU experiment behind DESIGN.md section 4: how coarse may the GEMM operands of the mel ResUNet be?

The oracle's functional ResUNet (oracle/resunet.py, torch CPU) is run in float64 with the operands of every
convolution (activated input and weight) rounded to a 16-bit format, fp32 accumulation emulated by the float64 sum, and
compared with the unrounded float64 forward: log-mel L1 and max error (bar: L1 <= 1e-3).  Formats: "split" = bf16 hi+lo
(what the kernels compute with three MFMAs), "bf16", "fp16" (one MFMA).  A per-level assignment keeps the levels listed in
--split on the split format and rounds the others to --fmt, to see which levels tolerate one MFMA per product.

    python scripts/operand_precision.py [--frames 320] [--fmt fp16] [--split 1,2]     (levels: 1 = full resolution ... 6, 7 = bottleneck)
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import resunet  # noqa: E402
from voicefixer_main_amd import synth  # noqa: E402


def rnd(x, fmt):
    if fmt == "exact":
        return x
    x32 = x.to(torch.float32)
    if fmt == "bf16":
        return x32.to(torch.bfloat16).to(x.dtype)
    if fmt == "fp16":
        return x32.clamp(-65504, 65504).to(torch.float16).to(x.dtype)
    if fmt == "split":
        hi = x32.to(torch.bfloat16).to(torch.float32)
        lo = (x32 - hi).to(torch.bfloat16).to(torch.float32)
        return (hi + lo).to(x.dtype)
    raise ValueError(fmt)


LEVEL_BY_CHANNELS = {32: 1, 64: 2, 128: 3, 256: 4, 384: 5}   # by output channels; 384 appears at levels 5, 6 and the bottleneck


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=320)
    ap.add_argument("--fmt", default="fp16")
    ap.add_argument("--split", default="", help="comma-separated levels kept on split-bf16 (1..5; 5 = all 384-channel levels)")
    args = ap.parse_args()
    keep = {int(v) for v in args.split.split(",") if v}
    sd = {k: v.double() for k, v in synth.make_resunet_state_dict(0).items()}
    rng = np.random.default_rng(0)
    mel = torch.from_numpy((10.0 ** (rng.normal(size=(1, 1, args.frames, 128)) * 1.2 - 2.5))).double()

    conv2d, convT = F.conv2d, F.conv_transpose2d
    mode = {"fmt": "exact"}

    def fmt_for(cout):
        if mode["fmt"] == "exact":
            return "exact"
        lvl = LEVEL_BY_CHANNELS.get(cout, 0)
        return "split" if (lvl in keep or lvl == 0) else mode["fmt"]

    def c2(x, w, *a, **k):
        f = fmt_for(w.shape[0]) if w.shape[1] > 1 else "exact"   # the Cin = 1 entry conv is not a GEMM kernel
        return conv2d(rnd(x, f), rnd(w, f), *a, **k)

    def ct(x, w, *a, **k):
        f = fmt_for(w.shape[1])
        return convT(rnd(x, f), rnd(w, f), *a, **k)

    ref = resunet.generator_mel(sd, mel)
    F.conv2d, F.conv_transpose2d = c2, ct
    try:
        for fmt in ("split", "bf16", "fp16") if not keep and args.fmt == "fp16" else (args.fmt,):
            mode["fmt"] = fmt
            out = resunet.generator_mel(sd, mel)
            d = (out - ref).abs()
            print("%-6s split levels %-10s log-mel L1 %.3e  max %.3e" % (fmt, sorted(keep) or "-", d.mean().item(), d.max().item()))
    finally:
        F.conv2d, F.conv_transpose2d = conv2d, convT


if __name__ == "__main__":
    main()
