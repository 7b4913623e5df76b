#!/usr/bin/env python3
"""Per-dispatch PMC report from the rocprofv3 databases written by scripts/pmc_passes.sh.

    python scripts/pmc_report.py gpurun_out/pmc_a [min_us]

Sums each counter over its instances per dispatch, keeps the conv launches of the last bench step and
prints one line per distinct (kernel, grid): duration, MFMA-busy %, wave-cycle split, LDS and HBM figures.
FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM section) prescribes for 16-byte-per-lane streaming reads.
"""
import collections
import glob
import re
import sqlite3
import sys


from kname import short  # noqa: E402  (scripts/ is on sys.path when run as a script)


def load(db):
    c = sqlite3.connect(db)
    rows = c.execute("select dispatch_id, kernel_name, grid_size, counter_name, sum(value), min(start), max(end) "
                     "from counters_collection group by dispatch_id, counter_name order by dispatch_id").fetchall()
    disp = collections.OrderedDict()
    for did, kn, grid, cn, val, st, en in rows:
        d = disp.setdefault(did, {"name": short(kn), "grid": grid, "us": (en - st) / 1e3})
        d[cn] = val
    return list(disp.values())


def main():
    root = sys.argv[1]
    min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
    passes = {}
    for db in sorted(glob.glob(root + "/*/*_results.db")):
        passes[db.split("/")[-2]] = load(db)
    base = passes[sorted(passes)[0]]
    n = len(base)
    merged = [dict(d) for d in base]
    for name, lst in passes.items():
        if len(lst) != n:
            print("pass %s has %d dispatches, expected %d -- skipped" % (name, len(lst), n))
            continue
        for m, d in zip(merged, lst):
            for k, v in d.items():
                if k not in ("name", "grid"):
                    m.setdefault(k, v)
            m["us_" + name] = d["us"]
    # last step = after the last k_stft_mel
    idx = [i for i, d in enumerate(merged) if d["name"].startswith("k_stft_mel")]
    step = merged[idx[-1]:] if idx else merged
    groups = collections.OrderedDict()
    for d in step:
        if d["us"] < min_us:
            continue
        groups.setdefault((d["name"], d["grid"]), []).append(d)
    hdr = ("kernel", "blocks", "n", "us", "mfma%", "valu%", "act%", "wait%", "winst%", "ldsconf%", "rdGB", "wrGB", "TB/s", "l2hit%", "GHz")
    print("%-30s %8s %3s %8s %6s %6s %6s %6s %6s %8s %7s %7s %6s %6s %5s" % hdr)
    for (name, grid), ds in groups.items():
        def avg(k):
            vals = [d[k] for d in ds if k in d]
            return sum(vals) / len(vals) if vals else float("nan")
        us = avg("us")
        wc = avg("SQ_WAVE_CYCLES")  # quad-cycles summed over waves
        busy = avg("SQ_BUSY_CYCLES")
        mfma = avg("SQ_VALU_MFMA_BUSY_CYCLES")
        gui = avg("GRBM_GUI_ACTIVE")
        # SQ_VALU_MFMA_BUSY_CYCLES sums over the 1024 SIMDs; GRBM_GUI_ACTIVE sums over the 8 XCDs
        cyc = gui / 8.0 if gui == gui else us * 2.0e3
        mfma_pct = 100.0 * mfma / (cyc * 1024.0) if mfma == mfma else float("nan")
        rd = 2.0 * avg("FETCH_SIZE") * 1024 / 1e9  # FETCH_SIZE is in KB
        wr = avg("WRITE_SIZE") * 1024 / 1e9
        hit, miss = avg("TCC_HIT_sum"), avg("TCC_MISS_sum")
        # effective shader clock of the counter pass that carried GRBM_GUI_ACTIVE: busy cycles per XCD / that pass's duration
        us_gui = avg("us_tcc1") if any("us_tcc1" in d for d in ds) else us
        ghz = gui / 8.0 / (us_gui * 1e3) if gui == gui else float("nan")
        print("%-30s %8d %3d %8.1f %6.1f %6.1f %6.1f %6.1f %6.1f %8.2f %7.3f %7.3f %6.2f %6.1f %5.2f" % (
            name, grid // 256, len(ds), us, mfma_pct,
            100.0 * avg("SQ_ACTIVE_INST_VALU") / wc, 100.0 * avg("SQ_ACTIVE_INST_ANY") / wc,
            100.0 * avg("SQ_WAIT_ANY") / wc, 100.0 * avg("SQ_WAIT_INST_ANY") / wc,
            100.0 * avg("SQ_LDS_BANK_CONFLICT") / max(avg("SQ_LDS_IDX_ACTIVE"), 1.0),
            rd, wr, (rd + wr) / (us * 1e-6) / 1e3, 100.0 * hit / max(hit + miss, 1.0), ghz))


if __name__ == "__main__":
    main()
