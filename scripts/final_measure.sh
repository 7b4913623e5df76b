#!/bin/bash
# One GPU call: GPU tests, smoke(), the default bench line (with aux workloads, power / clock sampler, live PMC traffic), the
# sharded workload at N = 1, the handler path, a kernel trace of the default bench and the PMC passes.
#   scripts/final_measure.sh <tag>      ->  gpurun_out/<tag>/{gpu_tests.log, bench_*.json, kernel_stats.*, pmc_report.txt, handler_*.json}
tag=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/$tag
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -n 3 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -n 1 $O/smoke.log
bash scripts/smi_sample.sh $O/smi.txt timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_gsr16x10.json 2> $O/bench_gsr16x10.err; cut -c1-200 $O/bench_gsr16x10.json
timeout 300 python bench.py --workload sharded1024 --steps 3 --warmup 1 --no-aux > $O/bench_sharded1024.json 2> $O/bench_sharded1024.err; cut -c1-160 $O/bench_sharded1024.json
timeout 200 python scripts/bench_handler.py --precision=2 > $O/handler_p2.json 2> $O/handler.err; cat $O/handler_p2.json | cut -c1-300
( cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats -d "$ROOT/$O/prof" -o $tag -- \
    python "$ROOT/bench.py" --steps 5 --warmup 2 --no-alt --no-aux --cpu-baseline-clips 0 --traffic off --no-parity > "$ROOT/$O/prof.log" 2>&1; echo "prof rc=$?" )
python scripts/prof_steps.py $(ls $O/prof/*/*_results.db $O/prof/*_results.db 2>/dev/null | head -1) --csv $O/kernel_stats.csv > $O/kernel_stats.txt 2>&1; head -n 14 $O/kernel_stats.txt
bash scripts/pmc_passes.sh $O/pmc --precision 2
python scripts/pmc_report.py $O/pmc 150 > $O/pmc_report.txt 2>&1; head -n 40 $O/pmc_report.txt
rm -rf $O/pmc/*/*.db $O/pmc/*/*/*.db $O/prof   # the databases are large; the reports stay
grep -c . $O/smi.txt; grep "GPU use (%): 100" $O/smi.txt | tail -n 3 | cut -c1-400
ls $O
