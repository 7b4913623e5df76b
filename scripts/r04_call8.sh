#!/bin/bash
# Round 4, GPU call 8: split-K of the deep ResUNet launches -- more slices (fewer stages per slice, more blocks per clip) against
# the round-2 rule (<= 8 slices, >= 3 stages per slice, 128 blocks per clip): mel ResUNet at the benched shape, the 1-s streaming
# chunk, the ssr batch.
O=gpurun_out/r04c8
mkdir -p $O
for v in default ks_a ks_b ks_c default; do
  if [ $v = default ]; then unset VFX_LIB_PATH; else export VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_$v.so; fi
  timeout 120 python scripts/unet_time.py $v --reps=10 --json=$O/unet.jsonl 2>&1 | grep "=="
  timeout 200 python bench.py --workload stream1s --steps 100 --warmup 10 --no-roofline --no-parity --cpu-baseline-clips 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('   stream1s', d['ms_per_step'], 'ms')"
  timeout 200 python bench.py --workload ssr_sr64 --steps 3 --warmup 1 --no-roofline --no-parity --cpu-baseline-clips 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('   ssr_sr64', d['ms_per_step'], 'ms')"
done
ls $O
