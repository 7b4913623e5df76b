#!/bin/bash
# Round 4, GPU call 19: the split-bf16 launches of ResUNet levels 3-6 on 64-cout tiles (three blocks per CU, the patch staged twice)
# instead of 128-cout tiles (two blocks per CU: the tile needs ~200 VGPRs) -- variant library, -DVFX_ABL_SPLIT_MAX_BN=64.
O=gpurun_out/r04c19
mkdir -p $O
VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_splitbn64.so timeout 200 python -m pytest tests/test_gpu_surface.py -m gpu -x -q -k "golden" > $O/tests_variant.log 2>&1; tail -n 2 $O/tests_variant.log
for v in default splitbn64 default splitbn64; do
  if [ $v = default ]; then unset VFX_LIB_PATH; else export VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_$v.so; fi
  timeout 100 python scripts/unet_time.py $v --reps=10 --json=$O/unet.jsonl 2>&1 | grep "=="
done
