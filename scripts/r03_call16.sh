#!/bin/bash
# Round 3, GPU call 16: rocprofv3 kernel trace and PMC passes of the final commit (the r03f ones predate the C = 128 pair).
tag=r03i
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/$tag
mkdir -p $O
timeout 300 python bench.py --steps 20 --warmup 5 --no-aux --no-alt --cpu-baseline-clips 0 > $O/bench_gsr16x10_short.json 2> $O/bench.err; cut -c1-160 $O/bench_gsr16x10_short.json
( cd /tmp; export TMPDIR=/tmp; timeout 200 rocprofv3 --kernel-trace --stats -d "$ROOT/$O/prof" -o $tag -- \
    python "$ROOT/bench.py" --steps 5 --warmup 2 --no-alt --no-aux --cpu-baseline-clips 0 --traffic off --no-parity > "$ROOT/$O/prof.log" 2>&1; echo "prof rc=$?" )
python scripts/prof_steps.py $(ls $O/prof/*/*_results.db $O/prof/*_results.db 2>/dev/null | head -1) --csv $O/kernel_stats.csv > $O/kernel_stats.txt 2>&1; head -n 8 $O/kernel_stats.txt
bash scripts/pmc_passes.sh $O/pmc --precision 2
python scripts/pmc_report.py $O/pmc 150 > $O/pmc_report.txt 2>&1; grep "k_resblock<256\|k_resblock<128\|k_resblock_pair<128" $O/pmc_report.txt | head -12
rm -rf $O/pmc/*/*.db $O/pmc/*/*/*.db $O/prof
ls $O
