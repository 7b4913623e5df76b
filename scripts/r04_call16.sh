#!/bin/bash
# Round 4, GPU call 16: the mel ResUNet's upsamplers (odd output width 2 W + 1) as TWO phased launches instead of four
# (TapConvParams::out_cmul) against a variant library without it (-DVFX_ABL_NO_ODD_PHASED on resunet.cpp).
O=gpurun_out/r04c16
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_surface.py tests/test_gpu_models.py -m gpu -x -q -k "golden or resunet or tuning or restore_gsr or poison or sub_batches" > $O/tests_models.log 2>&1; tail -n 4 $O/tests_models.log
timeout 600 python -m pytest tests/test_gpu_shapes.py -m gpu -x -q -k "benched_shape" > $O/tests_shapes.log 2>&1; tail -n 3 $O/tests_shapes.log
for v in default nooddphase default nooddphase; do
  if [ $v = default ]; then unset VFX_LIB_PATH; else export VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_$v.so; fi
  timeout 120 python scripts/unet_time.py $v --reps=10 --json=$O/unet.jsonl 2>&1 | grep "=="
done
for v in default nooddphase default nooddphase; do
  if [ $v = default ]; then unset VFX_LIB_PATH; else export VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_$v.so; fi
  timeout 200 python bench.py --workload stream1s --steps 100 --warmup 10 --no-roofline --no-parity --cpu-baseline-clips 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('stream1s $v', d['ms_per_step'], 'ms')" | tee -a $O/stream.txt
done
