#!/bin/bash
# Round 4, GPU call 13: the 16 x 16 tile at C = 64 as ONE 8-wave block per CU (variant library, -DVFX_RB_C64_T256) against the product.
O=gpurun_out/r04c13
mkdir -p $O
VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_c64t256.so timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_surface.py -m gpu -x -q -k "fused_conv_block or golden" > $O/tests_variant.log 2>&1; tail -n 3 $O/tests_variant.log
for v in default c64t256 default c64t256; do
  if [ $v = default ]; then unset VFX_LIB_PATH; else export VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_$v.so; fi
  timeout 120 python scripts/unet_time.py $v --reps=10 --json=$O/unet.jsonl 2>&1 | grep "=="
done
for v in default c64t256; do
  if [ $v = default ]; then unset VFX_LIB_PATH; else export VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_$v.so; fi
  timeout 300 python bench.py --workload ssr_sr64 --steps 3 --warmup 1 --no-roofline --no-parity --cpu-baseline-clips 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ssr_sr64 $v', d['ms_per_step'], 'ms')" | tee -a $O/ssr.txt
  timeout 200 python bench.py --workload stream1s --steps 100 --warmup 10 --no-roofline --no-parity --cpu-baseline-clips 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('stream1s $v', d['ms_per_step'], 'ms')" | tee -a $O/ssr.txt
done
