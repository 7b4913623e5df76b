#!/usr/bin/env python3
"""Static check of the hand-counted weight loads in the convolution kernels' gfx950 assembly.

    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only conv.hip -o conv.s;  python scripts/asm_inflight_check.py conv.s

The inline-asm `global_load_dwordx4` of load_b_asm() is invisible to the compiler's s_waitcnt bookkeeping: until the
hand-written `s_waitcnt vmcnt(N)` that covers it, its destination registers hold stale values.  A compiler-generated
copy (v_mov) or any other read of them in that window is a silent bug that only shows when the registers hold
garbage (first launch on a device).  The scan walks every kernel in layout order, counts vector-memory operations,
retires a load at the first `s_waitcnt vmcnt(N)` with N < (operations issued after it), and reports every other
instruction that touches an in-flight destination before that.  MFMAs count as touches too.
"""
import re
import sys


def regs(tok):
    out = set()
    for m in re.finditer(r'v\[(\d+):(\d+)\]', tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\bv(\d+)\b', tok):
        out.add(int(m.group(1)))
    return out


def main():
    src = open(sys.argv[1]).read()
    total = 0
    for fn in re.finditer(r'\n(_ZN3vfx[A-Za-z0-9_]*):[^\n]*\n(.*?)s_endpgm', src, re.S):
        name, body = fn.group(1), fn.group(2).split('\n')
        inflight = []  # (regs, seq)
        seq = 0
        bad = 0
        inasm = False
        for i, l in enumerate(body):
            if 'ASMSTART' in l:
                inasm = True
            if 'ASMEND' in l:
                inasm = False
            c = l.split(';')[0].strip()
            if not c or c.endswith(':') or c.startswith('.'):
                continue
            op = c.split()[0]
            m = re.match(r's_waitcnt.*vmcnt\((\d+)\)', c)
            if m:
                n = int(m.group(1))
                # at most n operations remain outstanding = the n youngest; a load with (seq - s) >= n younger ones has landed
                inflight = [(r, s) for r, s in inflight if seq - s < n]
                continue
            is_vmem = op.startswith(('global_load', 'global_store', 'buffer_load', 'buffer_store', 'global_atomic', 'scratch_'))
            if is_vmem:
                seq += 1
                if inasm and op.startswith('global_load_dwordx4'):
                    inflight.append((regs(c.split(',')[0]), seq))
                    continue
            touched = regs(c)
            for r, s in inflight:
                if touched & r:
                    bad += 1
                    if bad <= 4:
                        print('   %s  line %d: %s' % (name[:70], i, c))
                    break
        if bad:
            print('%s: %d instruction(s) touch in-flight weight registers' % (name, bad))
        total += bad
    print('total', total)
    return 1 if total else 0


if __name__ == '__main__':
    sys.exit(main())
