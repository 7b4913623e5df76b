#!/usr/bin/env python3
"""LDS bank-conflict model for the access patterns of csrc/stft.hip's wave-level FFT (MI355X_MICROARCH.md, LDS section):
64 banks of 4 bytes; a ds_read_b64 / ds_write_b64 is served in two groups of 32 lanes ({0-31}, {32-63}), a lane touching two
consecutive banks; lanes of a group conflict when they touch the same bank at different addresses (cost = max distinct
addresses on one bank).  Prints the cost (1 = conflict-free) of every access of the three passes and of the untangle.

    python scripts/lds_conflicts.py
"""
import collections


def cost_b64(addr_of_lane):
    worst = 1
    for grp in (range(0, 32), range(32, 64)):
        banks = collections.defaultdict(set)
        for l in grp:
            a = addr_of_lane(l)          # index in float2 units -> byte address 8 a
            for b in (2 * a, 2 * a + 1):  # the two dwords
                banks[b % 64].add(b)
        worst = max(worst, max(len(v) for v in banks.values()))
    return worst


def main():
    pad = lambda i: i + (i >> 4)
    rows = []
    rows.append(("pass 1 write  17 lane + r", max(cost_b64(lambda l, r=r: 17 * l + r) for r in range(16))))
    rows.append(("pass 2 read   lane + lane/16 + 68 r", max(cost_b64(lambda l, r=r: l + (l >> 4) + 68 * r) for r in range(16))))
    rows.append(("pass 2 write  17 (lane - k) + k + 17 r",
                 max(cost_b64(lambda l, r=r: 17 * (l - (l & 15)) + (l & 15) + 17 * r) for r in range(16))))
    rows.append(("pass 3 read   lane + lane/16 + 68 q + 272 r",
                 max(cost_b64(lambda l, q=q, r=r: l + (l >> 4) + 68 * q + 272 * r) for q in range(4) for r in range(4))))
    rows.append(("untangle write (natural order)", max(cost_b64(lambda l, m=m: pad(l + 64 * m)) for m in range(16))))
    rows.append(("untangle read  Z[1024 - k]", max(cost_b64(lambda l, m=m: pad((1024 - (l + 64 * m)) & 1023)) for m in range(16))))
    rows.append(("unpadded pass 1 write 16 lane + r (for comparison)", max(cost_b64(lambda l, r=r: 16 * l + r) for r in range(16))))
    for name, c in rows:
        print("%-52s %d-way" % (name, c))


if __name__ == "__main__":
    main()
