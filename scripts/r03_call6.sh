#!/bin/bash
# Round 3, GPU call 5: deep residual prefetch (w64), register-resident residual without LDS hand-off (r128), atomic handler output.
O=gpurun_out/r03c6
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "resblock" > $O/tests_resblock.log 2>&1; tail -3 $O/tests_resblock.log
timeout 1200 python -m pytest tests/test_gpu_surface.py tests/test_gpu_models.py -m gpu -x -q > $O/tests_models.log 2>&1; tail -3 $O/tests_models.log
for t in 0 256 192; do
  timeout 200 python scripts/voc_layers.py tuning_$t --tuning=$t --reps=5 --json=$O/ab.jsonl > $O/ab_tuning_$t.txt 2>&1
done
grep -h "==\|k_resblock<256\|k_resblock<128" $O/ab_tuning_*.txt | grep -v "d="
grep -h "d=" $O/ab_tuning_0.txt | head -16
VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_timing.so timeout 600 python scripts/phase_timing.py --json=$O/phase_timing.json > $O/phase_timing.txt 2>&1; cat $O/phase_timing.txt | grep -v amdgpu.ids
timeout 600 python bench.py --steps 20 --warmup 5 --no-aux > $O/bench_gsr16x10.json 2> $O/bench.err; cut -c1-200 $O/bench_gsr16x10.json; tail -2 $O/bench.err
ls $O
