#!/bin/bash
# Round 4, GPU call 21: even patch width for the upsampler launches (plan_conv) against a variant library that keeps the odd width
# (-DVFX_ABL_ODD_PATCH_WIDTH on api.cpp): transposed-convolution kernel tests, reference goldens, mel ResUNet A/B.
O=gpurun_out/r04c21
mkdir -p $O
timeout 100 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_surface.py -m gpu -x -q -k "conv_transpose2d or golden" > $O/tests.log 2>&1; tail -n 2 $O/tests.log
for v in default oddpw default oddpw; do
  if [ $v = default ]; then unset VFX_LIB_PATH; else export VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_$v.so; fi
  timeout 60 python scripts/unet_time.py $v --reps=10 --json=$O/unet.jsonl 2>&1 | grep "=="
done
