#!/usr/bin/env python3
"""Time one fused C = 32 ConvBlockRes launch at the level-1 shape of the benched batch (16 x 1016 x 128), 50 launches back to back
(VFX_OP_REPS in libvfx_test.so's vfx_op_block2d); --old: k_resblock's 16 x 16 form (VFX_TUNE_OLD_BLOCK2D).

    [VFX_LIB_PATH=...] python scripts/b2d_time.py [--old] [B H W]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VFX_OP_REPS"] = "50"
from voicefixer_main_amd import _lib  # noqa: E402
from voicefixer_main_amd.engine import Engine  # noqa: E402

B, H, W, C = 16, 1016, 128, 32
_shape = [int(a) for a in sys.argv[1:] if a.isdigit()]
if len(_shape) == 3:
    B, H, W = _shape
g = torch.Generator().manual_seed(3)
x = (torch.randn((B, H, W, C), generator=g) * 3.0).cuda()
w1, w2 = (torch.randn((C, C, 3, 3), generator=g) * 0.06).numpy(), (torch.randn((C, C, 3, 3), generator=g) * 0.06).numpy()
sc = (torch.rand(C, generator=g) + 0.5).numpy()
sh = (torch.randn(C, generator=g) * 0.2).numpy()
eng = Engine("cuda:0", config={"precision": 1, "tuning": _lib.TUNE_OLD_BLOCK2D if "--old" in sys.argv else 0})
print(os.environ.get("VFX_LIB_PATH", "libvfx.so"), "--old" if "--old" in sys.argv else "", (B, H, W), "tiles:", B * ((H + 13) // 14) * ((W + 13) // 14), flush=True)
for _ in range(2):
    eng.op_block2d(x, w1, sc, sh, w2, sc, sh, 0.01)
