#!/bin/bash
# Round 3, GPU call 2: the 4-wave / two-blocks-per-CU form of the C = 256 layer (resblock_w64.hip): parity, A/B, occupancy, PMC.
O=gpurun_out/r03c2
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "resblock" > $O/tests_resblock.log 2>&1; tail -3 $O/tests_resblock.log
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_shapes.py -m gpu -x -q > $O/tests_models.log 2>&1; tail -3 $O/tests_models.log
for v in 1 0; do
  VFX_RBA_W64=$v timeout 200 python scripts/voc_layers.py w64_$v --reps=5 --json=$O/ab.jsonl > $O/ab_w64_$v.txt 2>&1
done
grep -h "==\|k_resblock<256" $O/ab_w64_*.txt
( cd /tmp; export TMPDIR=/tmp; timeout 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof" -o w64 -- python "$GRAFT_REPO_ROOT/scripts/voc_layers.py" prof --reps=2 > "$GRAFT_REPO_ROOT/$O/prof.log" 2>&1; echo "prof rc=$?" )
python scripts/prof_steps.py $(ls $O/prof/*/*_results.db $O/prof/*_results.db 2>/dev/null | head -1) --csv $O/kernel_stats.csv 2>&1 | head -30 > $O/kernel_stats.txt; head -12 $O/kernel_stats.txt
bash scripts/pmc_passes.sh $O/pmc --precision 2
python scripts/pmc_report.py $O/pmc 150 > $O/pmc_report.txt 2>&1; grep "kernel\|k_resblock" $O/pmc_report.txt
rm -rf $O/pmc/*/*.db $O/pmc/*/*/*.db $O/prof
cp gpurun_out/parity_shapes.json $O/ 2>/dev/null
ls $O
