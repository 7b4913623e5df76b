#!/usr/bin/env python3
"""HBM traffic per launch of every convolution kernel, from the rocprofv3 PMC passes of scripts/pmc_passes.sh.

    python scripts/pmc_traffic.py gpurun_out/pmc_x profiles/r01_traffic.json [clips seconds precision]

bytes = 2 * FETCH_SIZE + WRITE_SIZE (both in KiB): on gfx950 FETCH_SIZE reports half of the bytes of a wide
streaming read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE matched the written tensor sizes 1:1 on this
workload.  Averages over the launches of the last bench step; bench.py reports the entry of its dominant kernel.
"""
import collections
import json
import re
import sqlite3
import sys


from kname import short  # noqa: E402  (scripts/ is on sys.path when run as a script)


def load(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select dispatch_id, kernel_name, sum(value) from counters_collection where counter_name=? "
                     "group by dispatch_id order by dispatch_id", (counter,)).fetchall()
    names = c.execute("select dispatch_id, kernel_name from counters_collection group by dispatch_id order by dispatch_id").fetchall()
    return [(short(k), v) for _, k, v in rows], [short(k) for _, k in names]


def main():
    root, out = sys.argv[1], sys.argv[2]
    clips, seconds, precision = (sys.argv[3:6] + ["16", "10", "1"])[:3] if len(sys.argv) > 3 else ("16", "10", "1")
    rd, names = load(root + "/tcc1/tcc1_results.db", "FETCH_SIZE")
    wr, _ = load(root + "/tcc2/tcc2_results.db", "WRITE_SIZE")
    last = max(i for i, n in enumerate(names) if n.startswith("k_stft_mel"))
    per = collections.OrderedDict()
    for (k, r), (_, w) in list(zip(rd, wr))[last:]:
        if not (k.startswith("k_conv") or k.startswith("k_resblock")):
            continue
        t = per.setdefault(k, [0, 0.0, 0.0])
        t[0] += 1
        t[1] += 2.0 * r * 1024.0
        t[2] += w * 1024.0
    res = {"workload": "%sx%ss" % (clips, seconds), "precision": int(precision),
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only); bytes = 2*FETCH_SIZE + "
                     "WRITE_SIZE per MI355X_MICROARCH.md; average per launch over one bench step",
           "kernels": {k: {"launches_per_step": v[0], "read_bytes_per_launch": round(v[1] / v[0]),
                           "write_bytes_per_launch": round(v[2] / v[0]), "bytes_per_launch": round((v[1] + v[2]) / v[0])}
                       for k, v in per.items()}}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
