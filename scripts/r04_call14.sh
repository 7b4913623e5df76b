#!/bin/bash
# Round 4, GPU call 14: the entry block (Cin = 1, k_resblock<.., IN1>) and the two-source block of decoder level 1 (SC2) as single
# launches, against variant libraries without one / both of them (-DVFX_ABL_NO_IN1 / -DVFX_ABL_NO_SC2 on resunet.cpp).
O=gpurun_out/r04c14
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_surface.py tests/test_gpu_models.py -m gpu -x -q -k "golden or resunet or tuning or restore_gsr or poison or sub_batches" > $O/tests_models.log 2>&1; tail -n 4 $O/tests_models.log
if ! grep -q " passed" $O/tests_models.log || grep -q "failed" $O/tests_models.log; then
  for v in noin1 nosc2; do
    VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_$v.so timeout 600 python -m pytest tests/test_gpu_surface.py tests/test_gpu_models.py -m gpu -x -q -k "golden or resunet_mel" > $O/tests_$v.log 2>&1; echo "variant $v:"; tail -n 3 $O/tests_$v.log
  done
fi
timeout 600 python -m pytest tests/test_gpu_shapes.py -m gpu -x -q > $O/tests_shapes.log 2>&1; tail -n 3 $O/tests_shapes.log
for v in default neither noin1 nosc2 default neither; do
  if [ $v = default ]; then unset VFX_LIB_PATH; else export VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_$v.so; fi
  timeout 120 python scripts/unet_time.py $v --reps=10 --json=$O/unet.jsonl 2>&1 | grep "=="
done
for v in default neither; do
  if [ $v = default ]; then unset VFX_LIB_PATH; else export VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_$v.so; fi
  timeout 300 python bench.py --workload ssr_sr64 --steps 3 --warmup 1 --no-roofline --no-parity --cpu-baseline-clips 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ssr_sr64 $v', d['ms_per_step'], 'ms')" | tee -a $O/ssr.txt
  timeout 200 python bench.py --workload stream1s --steps 100 --warmup 10 --no-roofline --no-parity --cpu-baseline-clips 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('stream1s $v', d['ms_per_step'], 'ms')" | tee -a $O/ssr.txt
done
unset VFX_LIB_PATH
VFX_PROFILE_DUMP=$O/convs_per_launch.csv timeout 300 python bench.py --steps 2 --warmup 1 --no-aux --no-alt --cpu-baseline-clips 0 --traffic off --no-parity > $O/bench_short.json 2> $O/bench_short.err; cut -c1-200 $O/bench_short.json
head -n 12 $O/convs_per_launch.csv
