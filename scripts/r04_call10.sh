#!/bin/bash
# Round 4, GPU call 10: the 1-s streaming chunk -- fused 2-D blocks against two launches per block (VFX_TUNE_NO_FUSED_UNET) and
# without split-K (VFX_TUNE_NO_SPLITK) at B = 1.
O=gpurun_out/r04c10
mkdir -p $O
for t in 0 4 32 0 4; do
  timeout 200 python bench.py --workload stream1s --tuning $t --steps 100 --warmup 10 --no-roofline --no-parity --cpu-baseline-clips 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('stream1s tuning $t', d['ms_per_step'], 'ms')" | tee -a $O/stream_tuning.txt
done
