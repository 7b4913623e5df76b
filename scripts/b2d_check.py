#!/usr/bin/env python3
"""Debug aid (round 6): the persistent C = 32 block kernel against k_resblock's 16 x 16 form on shapes with many tiles per block;
prints where they differ (image, tile row / column, position inside the tile).

    python scripts/b2d_check.py [B H W]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicefixer_main_amd import _lib  # noqa: E402
from voicefixer_main_amd.engine import Engine  # noqa: E402


def main():
    B, H, W = [int(a) for a in sys.argv[1:4]] if len(sys.argv) >= 4 else (16, 1016, 128)
    C = 32
    g = torch.Generator().manual_seed(3)
    x = (torch.randn((B, H, W, C), generator=g) * 3.0).cuda()
    w1, w2 = (torch.randn((C, C, 3, 3), generator=g) * 0.06).numpy(), (torch.randn((C, C, 3, 3), generator=g) * 0.06).numpy()
    sc = (torch.rand(C, generator=g) + 0.5).numpy()
    sh = (torch.randn(C, generator=g) * 0.2).numpy()
    new = Engine("cuda:0", config={"precision": 1})
    old = Engine("cuda:0", config={"precision": 1, "tuning": _lib.TUNE_OLD_BLOCK2D})
    ref = old.op_block2d(x, w1, sc, sh, w2, sc, sh, 0.01)
    for rep in range(4):
        y = new.op_block2d(x, w1, sc, sh, w2, sc, sh, 0.01)
        bad = (y != ref).any(dim=3)
        n = int(bad.sum().item())
        print("rep %d: %d of %d pixels differ" % (rep, n, B * H * W))
        if n:
            idx = bad.nonzero()[:4000].cpu().numpy()
            tiles = {}
            for b, i, j in idx:
                tiles.setdefault((int(b), int(i) // 14, int(j) // 14), []).append((int(i) % 14, int(j) % 14))
            for k in list(tiles)[:12]:
                v = tiles[k]
                print("   image %d tile (%d, %d): %d pixels, rows %s cols %s" % (k[0], k[1], k[2], len(v), sorted(set(a for a, _ in v)), sorted(set(c for _, c in v))))
            d = (y - ref).abs()
            print("   max abs difference %.3e" % d.max().item())
            ne = (y != ref)
            print("   differing elements per channel:", ne.sum(dim=(0, 1, 2)).cpu().numpy().tolist())
            bi = ne.any(dim=3).nonzero()
            print("   grid row (i %% 14 + 1) histogram:", np.bincount((bi[:, 1].cpu().numpy() % 14) + 1, minlength=16).tolist())
            print("   grid col (j %% 14 + 1) histogram:", np.bincount((bi[:, 2].cpu().numpy() % 14) + 1, minlength=16).tolist())
            dd = (y - ref)[ne]
            xx = x[ne]
            print("   wrong - right == -x for %d of %d elements; wrong == x for %d; wrong == 0 for %d" % (
                int(((dd + xx).abs() < 1e-5).sum()), int(ne.sum()), int(((y[ne] - xx).abs() < 1e-6).sum()), int((y[ne] == 0).sum())))
            for b, i, j in idx[:6]:
                c = 16
                want = (y[b, i, j, c] - ref[b, i, j, c] + x[b, i, j, c]).item()
                hit = ((x[b, :, :, :] - want).abs() < 2e-5).nonzero()[:5].cpu().numpy().tolist()
                print("   pixel (%d, %d, %d) channel 16: the residual that explains the value is x at (row, col, channel):" % (b, i, j), hit,
                      " [tile origin row %d col %d]" % (14 * (int(i) // 14), 14 * (int(j) // 14)))
            for b, i, j in idx[:1]:
                print("   pixel", int(b), int(i), int(j), "\n     y  ", y[b, i, j, :6].cpu().numpy(), "\n     ref", ref[b, i, j, :6].cpu().numpy(),
                      "\n     x  ", x[b, i, j, :6].cpu().numpy(), "\n     channels differing:", int((y[b, i, j] != ref[b, i, j]).sum().item()))
                # is it another pixel's value?
                cand = (ref[b] - y[b, i, j]).abs().amax(dim=2)
                m = (cand == 0).nonzero()
                cand2 = ((ref[b] - x[b]) - (y[b, i, j] - x[b, i, j])).abs().amax(dim=2)
                print("     equals ref at:", m[:4].cpu().numpy().tolist(), "; conv part (y - x) equals the conv part of:", (cand2 < 1e-6).nonzero()[:4].cpu().numpy().tolist())


if __name__ == "__main__":
    main()
