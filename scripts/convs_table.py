#!/usr/bin/env python3
"""Group the per-launch conv table written by VFX_PROFILE_DUMP (last step) by layer shape."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 213
last = rows[-n:]
prev, acc = None, []
for r in last:
    key = (r['M'], r['Cout'], r['K'], r['nseg'], r['ntaps0'], r['C0'], r['sw'])
    if prev and prev[0] == key:
        prev[1].append(float(r['ms']))
        prev[2].append(float(r['tflops']))
    else:
        prev = [key, [float(r['ms'])], [float(r['tflops'])]]
        acc.append(prev)
t = 0
for k, ms, tf in acc:
    t += sum(ms)
    print("M=%8s Cout=%4s K=%5s nseg=%s nt=%s C0=%4s sw=%s x%2d avg %7.1f us  %6.1f TF  cum %6.2f ms" % (
        *k, len(ms), 1e3 * sum(ms) / len(ms), sum(tf) / len(tf), t))
