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

RW_STAMPS = {0: "tile start", 1: "wait for the prefetched patch + patch rows -> LDS", 2: "barrier (patch complete)", 3: "request next patch",
             4: "conv1", 5: "residual read + h write", 6: "barrier (h complete)", 7: "conv2", 8: "y1 -> second patch + barrier",
             9: "conv1 (second layer)", 10: "h write (second layer)", 11: "barrier", 12: "conv2 (second layer)", 13: "epilogue (direct stores issued)"}


def rw_case(eng, buf, g, d, d2, out):
    """The persistent C = 64 kernel (resblock_rw.hip): 8 waves per block, stamps per tile; single layer or pair."""
    C, T, B = 64, 1006 * 441, 16
    x = torch.randn((B, T, C), generator=g).cuda()
    mk = lambda: ((torch.randn((C, C, 3), generator=g) * 0.05).numpy(), (torch.randn((C,), generator=g) * 0.1).numpy(),
                  (torch.randn((C, C, 3), generator=g) * 0.05).numpy(), (torch.randn((C,), generator=g) * 0.1).numpy())
    la, lb = mk(), mk()
    call = (lambda: eng.op_resblock(x, la[0], la[1], la[2], la[3], d, 0.01, True)) if d2 is None else \
        (lambda: eng.op_resblock_pair(x, la, d, lb, d2, 0.01))
    call()
    buf.zero_()
    torch.cuda.synchronize()
    call()
    torch.cuda.synchronize()
    ts = buf.cpu().numpy().astype(np.uint64).reshape(-1, 8, 16)
    ts = ts[ts[:, 0, 13] != 0].astype(np.float64)
    idx = [i for i in range(14) if ts[0, 0, i] != 0]
    life = ts[:, :, 13] - ts[:, :, 0]
    span = ts[:, :, 13].max() - ts[:, :, 0].min()
    name = "C64_d%d" % d + ("" if d2 is None else "_%d" % d2)
    res = {"tiles": int(ts.shape[0]), "mean_tile_cycles": float(life.mean()), "kernel_span_cycles": float(span), "phases": {}}
    print("== C = 64, d = %s: %d tiles, tile time %.0f cycles (mean over waves), kernel span %.0f cycles = %.1f tiles per CU back to back" % (
        d if d2 is None else (d, d2), ts.shape[0], life.mean(), span, span / life.mean()))
    for a, b in zip(idx[:-1], idx[1:]):
        v = float((ts[:, :, b] - ts[:, :, a]).mean())
        vmax = float((ts[:, :, b] - ts[:, :, a]).max(axis=1).mean())
        res["phases"]["%02d %s" % (b, RW_STAMPS[b])] = v
        print("   %-52s %8.0f cycles  %5.1f %%   (slowest wave of a tile: %6.0f)" % (RW_STAMPS[b], v, 100.0 * v / life.mean(), vmax))
    out[name] = res


def block2d_case(eng, buf, g, C, H, W, out, prec):
    """A fused 2-D ConvBlockRes of the ResUNets (resblock.hip, G2) at a level's shape: 4 waves per block, one tile per block."""
    B = 16
    x = torch.randn((B, H, W, C), generator=g).cuda()
    w1, w2 = (torch.randn((C, C, 3, 3), generator=g) * 0.05).numpy(), (torch.randn((C, C, 3, 3), generator=g) * 0.05).numpy()
    one, zero = np.ones(C, np.float32), np.zeros(C, np.float32)
    call = lambda: eng.op_block2d(x, w1, one, zero, w2, one, zero, 0.01)
    call()
    buf.zero_()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    call()
    ev1.record()
    torch.cuda.synchronize()
    ts = buf.cpu().numpy().astype(np.uint64).reshape(-1, 4, 16)
    ts = ts[ts[:, 0, 12] != 0].astype(np.float64)
    n = ts.shape[0]
    dt = np.diff(ts[:, :, :13], axis=2)
    life = ts[:, :, 12] - ts[:, :, 0]
    span = ts[:, :, 12].max() - ts[:, :, 0].min()
    res = {"tiles": int(n), "mean_block_cycles": float(life.mean()), "kernel_span_cycles": float(span),
           "phases": {("%02d %s" % (i, PHASES[i])): float(dt[:, :, i].mean()) for i in range(12)},
           "call_ms_incl_host_prep": ev0.elapsed_time(ev1)}
    out["block2d_C%d" % C] = res
    print("== 2-D block, C = %d, %d x %d x %d (precision %d): %d tiles, block lifetime %.0f cycles (mean), kernel span %.0f cycles, "
          "blocks in flight = %.1f per CU; the call: %.3f ms" % (C, B, H, W, prec, n, life.mean(), span, n * life.mean() / span / 256.0,
                                                                  res["call_ms_incl_host_prep"]))
    if C == 64 and float(ts[:, :, 13].max()) > 0:  # k_resblock<64, 4>, 2-D: conv1 of chunk 0 split further (stamps 13-15)
        c0 = (ts[:, :, 13] - ts[:, :, 4]).mean()
        w1_ = (ts[:, :, 14] - ts[:, :, 13]).mean()
        t1_ = (ts[:, :, 15] - ts[:, :, 14]).mean()
        c1 = (ts[:, :, 5] - ts[:, :, 15]).mean()
        print("   conv1 split: taps of chunk 0 %.0f cycles (incl. the request of chunk 1), wait for chunk 1 %.0f, transform of chunk 1 %.0f, "
              "barrier + taps of chunk 1 + final wait %.0f" % (c0, w1_, t1_, c1))
        res["conv1_split"] = {"chunk0_taps": float(c0), "chunk1_wait": float(w1_), "chunk1_transform": float(t1_), "chunk1_taps": float(c1)}
    # per-CU concurrency from the stamps themselves (s_memtime is not synchronised across XCDs: use each block's own start / end and
    # the kernel's duration): sum of tile lifetimes / (256 CUs x the call's duration in cycles at the clock the stamps imply)
    if C == 32:  # block2d32.hip: [start, end] of every block on the chip-wide 100 MHz clock + HW_ID / XCC_ID
        raw = buf.cpu().numpy().astype(np.uint64)
        nt = 16 * ((H + 13) // 14) * ((W + 13) // 14)
        tb = raw[nt * 64: nt * 64 + 768 * 4].reshape(768, 4)
        tb = tb[tb[:, 1] != 0]
        if tb.shape[0]:
            t0 = tb[:, 0].astype(np.float64) - float(tb[:, 0].min())
            t1 = tb[:, 1].astype(np.float64) - float(tb[:, 0].min())
            mid = 0.5 * float(t1.max())
            hw = tb[:, 2] & np.uint64(0xffffffff)
            xcc = (tb[:, 2] >> np.uint64(32)) & np.uint64(0xf)
            cu = (hw >> np.uint64(8)) & np.uint64(0xf)
            sh = (hw >> np.uint64(12)) & np.uint64(0x1)
            se = (hw >> np.uint64(13)) & np.uint64(0x7)
            key = xcc * np.uint64(1000) + se * np.uint64(100) + sh * np.uint64(20) + cu
            print("   blocks: %d; kernel %.1f us on the 100 MHz clock; starts: %d within 5 us, %d later; running at mid-kernel: %d; "
                  "distinct (XCC, SE, SH, CU): %d; blocks per CU: max %d" % (
                      tb.shape[0], t1.max() / 100.0, int((t0 < 500).sum()), int((t0 >= 500).sum()), int(((t0 < mid) & (t1 > mid)).sum()),
                      len(set(key.tolist())), max(np.bincount(np.unique(key, return_inverse=True)[1]))))
            late = np.sort(t0)[-8:] / 100.0
            print("   latest block starts (us):", " ".join("%.1f" % v for v in late), "; block durations (us): min %.1f mean %.1f max %.1f" % (
                (t1 - t0).min() / 100.0, (t1 - t0).mean() / 100.0, (t1 - t0).max() / 100.0))
    first, last = ts[:, :, 0].min(axis=1), ts[:, :, 12].max(axis=1)
    print("   sum of tile lifetimes %.3e cycles; mean tile %.0f; tiles x mean / 768 blocks = %.0f cycles per block" % (
        float((last - first).sum()), float((last - first).mean()), float((last - first).sum()) / 768.0))
    for i, (k, v) in enumerate(res["phases"].items()):
        print("   %-22s %9.0f cycles  %5.1f %%   (slowest wave of a block: %7.0f)" % (k, v, 100.0 * v / life.mean(), dt[:, :, i].max(axis=1).mean()))


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
    only2d = "--only-block2d" in sys.argv
    if "--block2d" in sys.argv or only2d:
        eng1 = Engine("cuda:0", config={"precision": 1})
        block2d_case(eng1, buf, g, 32, 1016, 128, out, 1)
        block2d_case(eng1, buf, g, 64, 504, 64, out, 1)
    if only2d:
        pass
    elif "--c64" in sys.argv or "--only-c64" in sys.argv:
        for d, d2 in ((1, None), (81, None), (243, None), (2187, None), (1, 3), (9, 27)):
            rw_case(eng, buf, g, d, d2, out)
    for C, T in (() if ("--only-c64" in sys.argv or only2d) else ((256, 49294), (128, 147882))):
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
