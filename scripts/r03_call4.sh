#!/bin/bash
# Round 3, GPU call 4: per-phase cycle breakdown of the 4-wave kernels, the full GPU suite, the default bench line.
O=gpurun_out/r03c4
mkdir -p $O
VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_timing.so timeout 600 python scripts/phase_timing.py --json=$O/phase_timing.json > $O/phase_timing.txt 2>&1; cat $O/phase_timing.txt | grep -v amdgpu.ids
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_gsr16x10.json 2> $O/bench.err; cut -c1-250 $O/bench_gsr16x10.json; tail -2 $O/bench.err
ls $O
