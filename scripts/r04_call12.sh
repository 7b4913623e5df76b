#!/bin/bash
# Round 4, GPU call 12: 16 x 16 h tiles for the fused C = 32 ConvBlockRes (k_resblock<32, 4, .., MT = 256>) against the 128-position
# tiles (VFX_TUNE_SMALL_2D_TILES): kernel / golden / model tests, mel ResUNet at the benched shape, the ssr batch.
O=gpurun_out/r04c12
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "fused_conv_block or conv3x3 or conv_transpose2d" > $O/tests_kernels.log 2>&1; tail -n 4 $O/tests_kernels.log
timeout 900 python -m pytest tests/test_gpu_surface.py tests/test_gpu_models.py -m gpu -x -q -k "golden or resunet or tuning or restore_gsr or poison" > $O/tests_models.log 2>&1; tail -n 4 $O/tests_models.log
for t in 128 0 128 0; do
  timeout 120 python scripts/unet_time.py tuning_$t --tuning=$t --reps=10 --json=$O/unet.jsonl 2>&1 | grep "=="
done
for t in 128 0; do
  timeout 300 python bench.py --workload ssr_sr64 --tuning $t --steps 3 --warmup 1 --no-roofline --no-parity --cpu-baseline-clips 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ssr_sr64 tuning $t', d['ms_per_step'], 'ms')" | tee -a $O/ssr.txt
  timeout 200 python bench.py --workload stream1s --tuning $t --steps 100 --warmup 10 --no-roofline --no-parity --cpu-baseline-clips 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('stream1s tuning $t', d['ms_per_step'], 'ms')" | tee -a $O/ssr.txt
done
ls $O
