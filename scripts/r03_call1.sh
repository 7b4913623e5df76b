#!/bin/bash
# Round 3, GPU call 1: GPU suite, stagger A/B of the two-blocks-per-CU ResStack kernels, the new bench line, the handler path.
O=gpurun_out/r03c1
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log
for s in 0 2 4 8; do
  VFX_RB_STAGGER=$s timeout 200 python scripts/voc_layers.py stagger$s --reps=5 --json=$O/stagger.jsonl > $O/stagger_$s.txt 2>&1
done
for s in 0 3 6; do
  VFX_RBA_MT=64 VFX_RB_STAGGER=$s timeout 200 python scripts/voc_layers.py mt64_stagger$s --reps=5 --json=$O/stagger.jsonl > $O/stagger_mt64_$s.txt 2>&1
done
grep -h "==\|k_resblock<128\|k_resblock<256" $O/stagger_*.txt | grep -v "d=" 
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_gsr16x10.json 2> $O/bench_gsr16x10.err; cut -c1-300 $O/bench_gsr16x10.json; tail -3 $O/bench_gsr16x10.err
timeout 200 python scripts/bench_handler.py --precision=1 > $O/handler_p1.json 2> $O/handler.err
timeout 200 python scripts/bench_handler.py --precision=2 > $O/handler_p2.json 2>> $O/handler.err
cat $O/handler_p1.json $O/handler_p2.json; tail -3 $O/handler.err
cp gpurun_out/parity_shapes.json gpurun_out/parity_prec2.json gpurun_out/fp16_robustness.json $O/ 2>/dev/null
ls $O
