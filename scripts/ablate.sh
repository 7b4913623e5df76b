#!/bin/bash
# Timing-only ablations of k_conv<128,false,true> (library built with EXTRA=-DVFX_ABLATION_BUILD).
mkdir -p gpurun_out/abl
for a in ${ABLS:-0 1 2 4 8 16 32 7 23 55}; do
  VFX_ABLATE=$a VFX_PROFILE_DUMP=gpurun_out/abl/convs_$a.csv timeout 200 python bench.py --cpu-baseline-clips 0 --steps 2 --warmup 1 > gpurun_out/abl/bench_$a.json 2> gpurun_out/abl/bench_$a.err
  echo "abl=$a rc=$?"
done
