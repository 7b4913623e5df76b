#!/usr/bin/env python3
"""CPU emulation of the INDEX LOGIC of the fused 2-D ConvBlockRes (k_resblock, G2 mode; resblock.hip) -- tile geometry of
plan_block2d, patch gather with zero fill, conv1 rows = arow1 + poff9, h validity, conv2 rows = clamp(m + hoff9), output
table -- with the arithmetic in float64 numpy, against torch.  It checks what a GPU run would otherwise have to debug
first; the MFMA / LDS layouts are shared with the verified 1-D mode.

    python scripts/emulate_block2d.py
"""
import numpy as np
import torch
import torch.nn.functional as F


def plan(H, W):
    best, W1, TH = -1.0, 0, 0
    for w1 in (16, 8):
        th = 128 // w1
        oh, ow = th - 2, w1 - 2
        covered = ((H + oh - 1) // oh) * oh * ((W + ow - 1) // ow) * ow
        util = H * W / covered
        if util > best:
            best, W1, TH = util, w1, th
    g = dict(W1=W1, TH=TH, TWo=W1 - 2, PW=W1 + 2)
    g["P"] = (TH + 2) * g["PW"]
    g["tiles_h"] = (H + TH - 3) // (TH - 2)
    g["tiles_w"] = (W + g["TWo"] - 1) // g["TWo"]
    g["poff9"] = [dy * g["PW"] + dx for dy in range(3) for dx in range(3)]
    g["hoff9"] = [(dy - 1) * W1 + (dx - 1) for dy in range(3) for dx in range(3)]
    assert g["P"] <= 192 and TH * W1 == 128
    return g


def emulate(x, w1, sc1, sh1, w2, sc2, sh2, slope):
    """x (B, H, W, C) channels-last float64; w (C, C, 3, 3)."""
    B, H, W, C = x.shape
    g = plan(H, W)
    W1, TH, PW, P = g["W1"], g["TH"], g["PW"], g["P"]
    y = np.full_like(x, np.nan)
    lrelu = lambda t: np.maximum(t, t * slope)
    wt1 = [w1[:, :, k // 3, k % 3] for k in range(9)]   # [cout, cin] per tap
    wt2 = [w2[:, :, k // 3, k % 3] for k in range(9)]
    for img in range(B):
        for ti in range(g["tiles_h"]):
            for tj in range(g["tiles_w"]):
                i0, j0 = ti * (TH - 2), tj * g["TWo"]
                # patch: rows 0..191, pixel (pi, pj) = image (i0 - 2 + pi, j0 - 2 + pj), activated, zero outside / beyond P
                patch = np.zeros((192, C))
                for prow in range(192):
                    pi, pj = divmod(prow, PW)
                    r, c = i0 - 2 + pi, j0 - 2 + pj
                    if prow < P and 0 <= r < H and 0 <= c < W:
                        patch[prow] = lrelu(x[img, r, c] * sc1 + sh1)
                # conv1 on the 128 h pixels
                h = np.zeros((128, C))
                for m in range(128):
                    li, lj = divmod(m, W1)
                    arow1 = li * PW + lj
                    acc = np.zeros(C)
                    for k in range(9):
                        acc += wt1[k] @ patch[arow1 + g["poff9"][k]]
                    r, c = i0 - 1 + li, j0 - 1 + lj
                    hval = li < TH and 0 <= r < H and 0 <= c < W
                    h[m] = lrelu(acc * sc2 + sh2) if hval else 0.0
                # conv2 on all 128 rows with clamped neighbours, masked store
                for m in range(128):
                    li, lj = divmod(m, W1)
                    r, c = i0 - 1 + li, j0 - 1 + lj
                    ok = 1 <= li <= TH - 2 and 1 <= lj <= W1 - 2 and r < H and c < W
                    if not ok:
                        continue
                    acc = np.zeros(C)
                    for k in range(9):
                        acc += wt2[k] @ h[min(max(m + g["hoff9"][k], 0), 127)]
                    assert np.isnan(y[img, r, c]).all(), "pixel written twice"
                    y[img, r, c] = acc + x[img, r, c]
    assert not np.isnan(y).any(), "pixel never written"
    return y


def main():
    rng = np.random.default_rng(0)
    for C, H, W in ((4, 40, 31), (3, 7, 5), (4, 13, 20), (2, 12, 14), (2, 6, 6)):
        B = 2
        x = rng.normal(size=(B, H, W, C))
        w1, w2 = rng.normal(size=(C, C, 3, 3)) * 0.3, rng.normal(size=(C, C, 3, 3)) * 0.3
        sc1, sc2 = rng.uniform(0.5, 1.5, C), rng.uniform(0.5, 1.5, C)
        sh1, sh2 = rng.normal(size=C) * 0.2, rng.normal(size=C) * 0.2
        got = emulate(x, w1, sc1, sh1, w2, sc2, sh2, 0.01)
        xt = torch.from_numpy(x).permute(0, 3, 1, 2)
        aff = lambda t, sc, sh: t * torch.from_numpy(sc)[None, :, None, None] + torch.from_numpy(sh)[None, :, None, None]
        hh = F.conv2d(F.leaky_relu(aff(xt, sc1, sh1), 0.01), torch.from_numpy(w1), padding=1)
        ref = (F.conv2d(F.leaky_relu(aff(hh, sc2, sh2), 0.01), torch.from_numpy(w2), padding=1) + xt).permute(0, 2, 3, 1).numpy()
        print("C=%d H=%d W=%d plan=%s  max err %.2e" % (C, H, W, {k: plan(H, W)[k] for k in ("W1", "TH", "tiles_h", "tiles_w")},
                                                       np.abs(got - ref).max()))
        assert np.abs(got - ref).max() < 1e-10


if __name__ == "__main__":
    main()
