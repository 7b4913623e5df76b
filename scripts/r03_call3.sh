#!/bin/bash
# Round 3, GPU call 3: the read-x-once 4-wave form of the C = 128 layer (resblock_r128.hip), 14-wide folded tiles, tuning switches.
O=gpurun_out/r03c3
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "resblock" > $O/tests_resblock.log 2>&1; tail -3 $O/tests_resblock.log
timeout 1200 python -m pytest tests/test_gpu_models.py -m gpu -x -q > $O/tests_models.log 2>&1; tail -3 $O/tests_models.log
for t in 0 128 64 192; do
  timeout 200 python scripts/voc_layers.py tuning_$t --tuning=$t --reps=5 --json=$O/ab.jsonl > $O/ab_tuning_$t.txt 2>&1
done
grep -h "==\|k_resblock<256\|k_resblock<128" $O/ab_tuning_*.txt | grep -v "d="
grep -h "d=" $O/ab_tuning_0.txt
bash scripts/pmc_passes.sh $O/pmc --precision 2
python scripts/pmc_report.py $O/pmc 150 > $O/pmc_report.txt 2>&1; grep "kernel\|k_resblock" $O/pmc_report.txt
rm -rf $O/pmc/*/*.db $O/pmc/*/*/*.db
timeout 600 python bench.py --steps 20 --warmup 5 --no-aux > $O/bench_gsr16x10.json 2> $O/bench.err; cut -c1-250 $O/bench_gsr16x10.json; tail -2 $O/bench.err
ls $O
