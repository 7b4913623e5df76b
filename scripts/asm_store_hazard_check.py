#!/usr/bin/env python3
"""Static check for the store-data hazard hipcc 7.2 leaves open on gfx950 (profiles/r06_store_data_hazard.md).

    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only block2d32.hip -o b.s;  python scripts/asm_store_hazard_check.py b.s

A buffer store of more than 8 bytes keeps reading its data registers after it has issued; a VALU instruction that writes one of them
must not be the very next instruction.  The compiler inserts the wait state when the store's soffset is an immediate, not when it is an
SGPR.  The scan reports every `buffer_store_dwordx3/x4 ..., sN offen|idxen|off` whose next instruction is a v_* that writes a data
register of the store (an s_nop, or anything else, in between is enough).
"""
import re
import sys


def vregs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]$', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


def main():
    src = open(sys.argv[1]).read()
    bad = 0
    for fn in re.finditer(r'\n(_ZN3vfx[A-Za-z0-9_]*):[^\n]*\n(.*?)s_endpgm', src, re.S):
        name = fn.group(1)
        insts = []
        for l in fn.group(2).split('\n'):
            c = l.split(';')[0].strip()
            if not c or c.endswith(':') or c.startswith('.'):
                continue
            insts.append(c)
        for i, c in enumerate(insts[:-1]):
            m = re.match(r'buffer_store_dwordx[34]\s+(v\[\d+:\d+\]),\s*[^,]+,\s*s\[\d+:\d+\],\s*(\S+)', c)
            if not m or not re.match(r's\d+$', m.group(2)):
                continue
            data = vregs(m.group(1))
            nxt = insts[i + 1]
            if not nxt.startswith('v_'):
                continue
            dst = nxt.split(None, 1)[1].split(',')[0].strip() if ' ' in nxt else ''
            if vregs(dst) & data:
                bad += 1
                print("%s: `%s` is followed at once by `%s`" % (name, c, nxt))
    print("%d unfenced store-data hazard(s)" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
