#!/usr/bin/env python3
"""tests/golden/chunker.npz from the REFERENCE's chunker classes (build container only).

    python oracle/gen_golden_chunker.py

Imports tools/dsp/overlapadd.py and tools/dsp/overlapadd_boxcar.py from /root/reference unmodified and runs
both `LambdaOverlapAdd` classes around a toy network (oracle.chunker.toy_nnet as a torch function).  The
window name is passed as "hann": the scipy in this image no longer knows the reference's default spelling
"hanning" (same window), and the rectangular cases pass "boxcar": `window=None` raises in the reference
constructors (overlapadd.py:411 calls `.type_as` on None), so the un-windowed branch is unreachable there.
Test infrastructure only.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)


class ToyNet(torch.nn.Module):
    """Carries the attribute chain the chunkers read for dtype (`nnet.f_helper.stft.conv_real.weight`,
    overlapadd.py:411)."""

    def __init__(self):
        super().__init__()
        self.f_helper = torch.nn.Module()
        self.f_helper.stft = torch.nn.Module()
        self.f_helper.stft.conv_real = torch.nn.Conv1d(1, 1, 1)

    def forward(self, x):
        prev = torch.cat([torch.zeros_like(x[..., :1]), x[..., :-1]], -1)
        n = x.shape[-1]
        ramp = (torch.arange(n, dtype=torch.float32) / n) * 0.05
        return {"wav": 0.6 * x + 0.3 * prev + ramp}


CASES_OLA = [  # (name, batch, n, window_size, hop_size, window)
    ("ola_hann_half", 2, 300, 64, None, "hann"),
    ("ola_rect_quarter", 1, 257, 64, 16, "boxcar"),
    ("ola_hann_short", 3, 40, 64, None, "hann"),
    ("ola_hann_exact", 1, 256, 64, 32, "hann"),
]
CASES_BOX = [  # (name, batch, n, window_size, in_margin, window)
    ("box_rect_ragged", 2, 300, 64, 8, "boxcar"),
    ("box_rect_exact", 1, 256, 64, 8, "boxcar"),
    ("box_hann_ragged", 1, 300, 64, 16, "hann"),
    ("box_rect_single", 2, 64, 64, 8, "boxcar"),
    ("box_rect_single_ragged", 1, 50, 64, 8, "boxcar"),
    ("box_rect_two", 1, 100, 64, 8, "boxcar"),
]


def main():
    from tools.dsp import overlapadd, overlapadd_boxcar
    net = ToyNet()
    out = {}
    rng = np.random.RandomState(1234)
    for name, B, n, W, hop, window in CASES_OLA:
        x = rng.uniform(-1, 1, (B, 1, n)).astype(np.float32)
        m = overlapadd.LambdaOverlapAdd(net, 1, W, hop_size=hop, window=window, reorder_chunks=True)
        y = m(torch.from_numpy(x))
        out[name + "_x"], out[name + "_y"] = x, y.numpy()
        out[name + "_cfg"] = np.array([W, hop if hop else W // 2, 1 if window == "hann" else 0])
    for name, B, n, W, M, window in CASES_BOX:
        x = rng.uniform(-1, 1, (B, 1, n)).astype(np.float32)
        # reorder_chunks=True raises in the boxcar class (zero overlap: overlapadd_boxcar.py:565-573 slices [-0:])
        m = overlapadd_boxcar.LambdaOverlapAdd(net, 1, W, M, window=window, reorder_chunks=False)
        y = m(torch.from_numpy(x))
        out[name + "_x"], out[name + "_y"] = x, y.numpy()
        out[name + "_cfg"] = np.array([W, M, 1 if window == "hann" else 0])
    path = os.path.join(ROOT, "tests", "golden", "chunker.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items() if k.endswith("_y")})


if __name__ == "__main__":
    main()
