#!/bin/bash
# Round 4, GPU call 3: the full GPU suite on the commit with the fp16 trunk, the simulator, the length-sorted sharding, the split
# test library; experiments (C = 64 as 4-wave blocks of 128 positions; residual a whole pass ahead at C = 256); phase stamps of the
# 4-wave kernels on the fp16 trunk; a fresh per-launch table.
O=gpurun_out/r04c3
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; tail -n 4 $O/gpu_tests.log
for t in 0 128; do
  timeout 200 python scripts/voc_layers.py tuning_$t --tuning=$t --reps=5 --json=$O/ab.jsonl > $O/ab_$t.txt 2>&1
done
VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_nrb4.so timeout 200 python scripts/voc_layers.py nrb4 --reps=5 --json=$O/ab.jsonl > $O/ab_nrb4.txt 2>&1
grep -h "==\|GEMM-shaped" $O/ab_*.txt
grep -h "k_resblock<64\|k_resblock_pair<64" $O/ab_0.txt $O/ab_128.txt
grep -h "k_resblock<256" $O/ab_0.txt $O/ab_nrb4.txt | grep -v "d="
VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_timing.so timeout 300 python scripts/phase_timing.py --json=$O/phase_timing.json > $O/phase_timing.txt 2>&1; cat $O/phase_timing.txt | tail -n 70
VFX_PROFILE_DUMP=$O/convs_per_launch.csv timeout 300 python bench.py --steps 2 --warmup 1 --no-aux --no-alt --cpu-baseline-clips 0 --traffic off --no-parity > $O/bench_short.json 2> $O/bench_short.err; cut -c1-200 $O/bench_short.json
ls $O
