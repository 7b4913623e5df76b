"""Kernel names as bench.py prints them, from the demangled names rocprofv3 records.

k_conv<BN, ELU, SPLIT, ABL, RING, HI>  ->  "k_conv<BN, ELU, SPLIT>" (+ " f16" for the 16-bit launches of precision 2)
k_resblock<C, NW, HI>                  ->  "k_resblock<C, NW>"      (+ " f16")
k_resblock_w64<C, X16>                 ->  "k_resblock<C, 4> f16"
k_resblock_r128<PAIR, X16>             ->  "k_resblock<128, 4> f16" / "k_resblock_pair<128, 4> f16"
k_resblock_rw<NW, PAIR, X16>           ->  "k_resblock<64, NW> f16" / "k_resblock_pair<64, NW> f16"
k_resblock_rw16<PAIR, HALO>            ->  "k_resblock<64, 8> f16" / "k_resblock_pair<64, 8> f16"
(X16 = the fp16 trunk of round 4: same rows of the tables, the bench line's `dtype` names the trunk format)
"""
import re


def short(n, width=40):
    m = re.search(r"vfx::(k_\w+)(?:<([^>]*)>)?", n)
    if not m:
        return n[:width]
    name, args = m.group(1), [a.strip() for a in (m.group(2) or "").split(",") if a.strip()]
    if name == "k_conv" and len(args) >= 3:
        hi = len(args) >= 6 and args[5] == "true"
        return "k_conv<%s, %s, %s>%s" % (args[0], args[1], args[2], " f16" if hi else "")
    if name == "k_resblock_w64" and args:                # the same layer as 4-wave blocks, two per CU
        return "k_resblock<%s, 4> f16" % args[0]
    if name == "k_resblock_r128":                        # C = 128, 16-bit mode: 4-wave blocks, x read once; <true>: a layer pair
        return "k_resblock_pair<128, 4> f16" if args and args[0] == "true" else "k_resblock<128, 4> f16"
    if name == "k_resblock_rw" and args:                 # C = 64, 16-bit mode: persistent, weights in registers
        if len(args) >= 2 and args[1] == "true":         # two layers per launch
            return "k_resblock_pair<64, %s> f16" % args[0]
        return "k_resblock<64, %s> f16" % args[0]
    if name == "k_resblock_rw16":                        # round 6: the same kernel with the lean tile loop (fp16 trunk); <PAIR, HALO>
        return "k_resblock_pair<64, 8> f16" if args and args[0] == "true" else "k_resblock<64, 8> f16"
    if name == "k_up16" and args:                        # round 6: the x3 ConvTranspose1d upsamplers of the 16-bit mode; <Cin / 64, TM>
        return "k_up16<%s> f16" % ", ".join(args)
    if name == "k_block2d32":                            # round 6: the persistent C = 32 ConvBlockRes of ResUNet level 1
        return "k_block2d<32, 4>"
    if name == "k_resblock" and len(args) >= 2:
        hi = len(args) >= 3 and args[2] == "true"
        return "k_resblock<%s, %s>%s" % (args[0], args[1], " f16" if hi else "")
    return name + ("<%s>" % ", ".join(args) if args else "")
