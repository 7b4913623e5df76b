#!/bin/bash
# Round 3, GPU call 7: 2-D swizzle key in the split BN = 128 tile (ResUNet levels 2-3), flag atomics through the global address
# space (the software-pipelined fragment reads of resblock_w64 / resblock_r128 were waited for with lgkmcnt(0)), 64 x 64 wave
# tiles in resblock_r128.  Libraries: abl/libvfx_head.so (before all three), abl/libvfx_base.so (+ key), abl/libvfx_atomfix_r32.so
# (+ key + atomics, 32 x 128 wave tiles), libvfx.so (everything).
O=gpurun_out/r03c7
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q > $O/tests_kernels.log 2>&1; tail -3 $O/tests_kernels.log
timeout 1200 python -m pytest tests/test_gpu_models.py -m gpu -x -q > $O/tests_models.log 2>&1; tail -3 $O/tests_models.log
for v in head atomfix_r32 default; do
  if [ $v = default ]; then unset VFX_LIB_PATH; else export VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_$v.so; fi
  timeout 200 python scripts/voc_layers.py lib_$v --reps=5 --json=$O/ab.jsonl > $O/ab_$v.txt 2>&1
done
unset VFX_LIB_PATH
grep -h "==\|k_resblock<256\|k_resblock<128" $O/ab_*.txt | grep -v "d="
grep -h "d=" $O/ab_default.txt | head -16
for v in head default; do
  if [ $v = default ]; then unset VFX_LIB_PATH; else export VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_$v.so; fi
  VFX_PROFILE_DUMP=$O/convs_$v.csv timeout 400 python bench.py --steps 10 --warmup 3 --no-aux --cpu-baseline-clips 0 --traffic off > $O/bench_gsr_$v.json 2> $O/bench_gsr_$v.err; cut -c1-160 $O/bench_gsr_$v.json
  timeout 400 python bench.py --workload ssr_sr64 --steps 5 --warmup 2 --no-aux --cpu-baseline-clips 0 --traffic off > $O/bench_ssr_$v.json 2> $O/bench_ssr_$v.err; cut -c1-160 $O/bench_ssr_$v.json
done
unset VFX_LIB_PATH
VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_timing.so timeout 600 python scripts/phase_timing.py --json=$O/phase_timing.json > $O/phase_timing.txt 2>&1; grep -v amdgpu.ids $O/phase_timing.txt
ls $O
