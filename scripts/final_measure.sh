#!/bin/bash
# One GPU call: GPU tests, the bench lines of every workload, a kernel-trace of the default bench and the PMC passes.
#   scripts/final_measure.sh <tag>      ->  gpurun_out/<tag>/{gpu_tests.log, bench_*.json, prof/, pmc/, handler_*.json}
tag=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/$tag
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
python bench.py --steps 20 --warmup 5 > $O/bench_gsr16x10.json 2> $O/bench_gsr16x10.err; cut -c1-200 $O/bench_gsr16x10.json
timeout 300 python bench.py --workload sharded1024 --steps 3 --warmup 1 > $O/bench_sharded1024.json 2> $O/bench_sharded1024.err
timeout 300 python bench.py --workload ssr_sr64 --steps 5 --warmup 2 > $O/bench_ssr_sr64.json 2> $O/bench_ssr_sr64.err
timeout 300 python bench.py --workload stream1s --steps 100 --warmup 10 > $O/bench_stream1s.json 2> $O/bench_stream1s.err
timeout 200 python scripts/bench_handler.py --precision=1 > $O/handler_p1.json 2> $O/handler.err
timeout 200 python scripts/bench_handler.py --precision=2 > $O/handler_p2.json 2>> $O/handler.err
( cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats -d "$ROOT/$O/prof" -o $tag -- \
    python "$ROOT/bench.py" --steps 5 --warmup 2 --no-alt --cpu-baseline-clips 0 --traffic off --no-parity > "$ROOT/$O/prof.log" 2>&1; echo "prof rc=$?" )
python scripts/prof_steps.py $(ls $O/prof/*/*_results.db $O/prof/*_results.db 2>/dev/null | head -1) --csv $O/kernel_stats.csv > $O/kernel_stats.txt 2>&1
bash scripts/pmc_passes.sh $O/pmc --precision 2
python scripts/pmc_report.py $O/pmc 150 > $O/pmc_report.txt 2>&1
rm -rf $O/pmc/*/*.db $O/prof   # the databases are large; the reports stay
ls $O
