#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel-trace database (rocpd .db): per-kernel totals and the launch
sequence of the last complete step (delimited by k_stft_mel launches), grouped by kernel + grid.

    python scripts/prof_steps.py gpurun_out/prof/x_results.db [--csv out.csv]
"""
import re
import sqlite3
import sys


from kname import short  # noqa: E402  (scripts/ is on sys.path when run as a script)


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    rows = list(c.execute("select name,start,end,grid_x,workgroup_x,lds_size,vgpr_count,accum_vgpr_count,sgpr_count "
                          "from kernels order by start"))
    tot = {}
    for r in rows:
        k = short(r[0])
        t = tot.setdefault(k, [0, 0.0, r[5], r[6], r[7], r[8]])
        t[0] += 1
        t[1] += (r[2] - r[1]) / 1e3
    all_us = sum(v[1] for v in tot.values())
    lines = ["kernel,calls,total_us,avg_us,pct,lds,vgpr,agpr,sgpr"]
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        lines.append("%s,%d,%.1f,%.2f,%.2f,%d,%d,%d,%d" % (k.replace(",", ";"), v[0], v[1], v[1] / v[0], 100 * v[1] / all_us,
                                                          v[2], v[3], v[4], v[5]))
    print("\n".join(lines))
    if "--csv" in sys.argv:
        open(sys.argv[sys.argv.index("--csv") + 1], "w").write("\n".join(lines) + "\n")
    idx = [i for i, r in enumerate(rows) if "k_stft_mel" in r[0]]
    if len(idx) < 2:
        return
    # (the run ends with a few front-end-only calls: take the last interval that holds a whole step, i.e. the longest one's length)
    longest = max(b - a for a, b in zip(idx, idx[1:]))
    lo, hi = [(a, b) for a, b in zip(idx, idx[1:]) if b - a == longest][-1]
    step = rows[lo:hi]
    print("\nlast full step: %d launches, span %.3f ms, kernel sum %.3f ms" % (
        len(step), (step[-1][2] - step[0][1]) / 1e6, sum(r[2] - r[1] for r in step) / 1e6))
    acc, prev = [], None
    for r in step:
        key = (short(r[0]), r[3] // r[4])
        d = (r[2] - r[1]) / 1e3
        if prev and prev[0] == key:
            prev[1].append(d)
        else:
            prev = [key, [d]]
            acc.append(prev)
    t = 0
    for k, ds in acc:
        t += sum(ds)
        print("%-36s grid %7d x%2d  avg %8.1f us  sum %8.1f  cum %8.1f" % (k[0], k[1], len(ds), sum(ds) / len(ds), sum(ds), t))


if __name__ == "__main__":
    main()
