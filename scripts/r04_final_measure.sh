#!/bin/bash
# Round 4: one GPU call with everything the judged numbers come from -- GPU tests, smoke(), the default bench line (aux workloads,
# power / clock sampler, live PMC traffic), r03 HEAD and r04 HEAD alternating on the SAME box, sharded workload at N = 1, the
# handler path, a kernel trace of the default bench and the PMC passes.
#   scripts/r04_final_measure.sh <tag>   ->  gpurun_out/<tag>/...
tag=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/$tag
mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -n 3 $O/smoke.log
# same-box A/B: the round-3 HEAD (its own tree and library under _ab_r03/) and this tree, alternating, identical flags
if [ -d _ab_r03 ]; then
  for i in 1 2; do
    ( cd _ab_r03 && timeout 300 python bench.py --steps 20 --warmup 5 --no-aux --no-alt --cpu-baseline-clips 0 --no-parity --traffic off --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); d['tree']='r03 HEAD (775c0d8)'; print(json.dumps(d))" ) >> $O/same_box_ab.jsonl
    timeout 300 python bench.py --steps 20 --warmup 5 --no-aux --no-alt --cpu-baseline-clips 0 --no-parity --traffic off --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); d['tree']='r04 HEAD'; print(json.dumps(d))" >> $O/same_box_ab.jsonl
  done
  python - <<P
import json
for l in open("$O/same_box_ab.jsonl"):
    d=json.loads(l); print(d["tree"], d["value"], d["ms_per_step"], d.get("ms_per_step_median"), (d.get("power") or {}).get("avg_sclk_mhz"))
P
fi
# (no rocm-smi poller beside the judged line: round 4 found it inflates the HIP-event average of the longest MFMA-dense kernel by 28 %)
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_gsr16x10.json 2> $O/bench_gsr16x10.err; cut -c1-200 $O/bench_gsr16x10.json; tail -n 2 $O/bench_gsr16x10.err
bash scripts/smi_sample.sh $O/smi.txt timeout 300 python bench.py --steps 20 --warmup 5 --no-aux --no-alt --cpu-baseline-clips 0 --no-parity --traffic off --no-roofline > /dev/null 2>&1
timeout 300 python bench.py --workload sharded1024 --steps 3 --warmup 1 --no-aux > $O/bench_sharded1024.json 2> $O/bench_sharded1024.err; cut -c1-160 $O/bench_sharded1024.json
timeout 300 python bench.py --workload ssr_sr64 --steps 3 --warmup 1 > $O/bench_ssr_sr64.json 2> $O/bench_ssr_sr64.err; cut -c1-160 $O/bench_ssr_sr64.json
timeout 300 python bench.py --workload stream1s --steps 100 --warmup 10 > $O/bench_stream1s.json 2> $O/bench_stream1s.err; cut -c1-160 $O/bench_stream1s.json
timeout 200 python scripts/bench_handler.py --precision=2 > $O/handler_p2.json 2> $O/handler.err; cat $O/handler_p2.json | cut -c1-300
( cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats -d "$ROOT/$O/prof" -o $tag -- \
    python "$ROOT/bench.py" --steps 5 --warmup 2 --no-alt --no-aux --cpu-baseline-clips 0 --traffic off --no-parity > "$ROOT/$O/prof.log" 2>&1; echo "prof rc=$?" )
python scripts/prof_steps.py $(ls $O/prof/*/*_results.db $O/prof/*_results.db 2>/dev/null | head -1) --csv $O/kernel_stats.csv > $O/kernel_stats.txt 2>&1; head -n 16 $O/kernel_stats.txt
( cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats -d "$ROOT/$O/prof_ssr" -o ${tag}_ssr -- \
    python "$ROOT/bench.py" --workload ssr_sr64 --steps 3 --warmup 1 --no-alt --no-aux --cpu-baseline-clips 0 --traffic off --no-parity --no-roofline > "$ROOT/$O/prof_ssr.log" 2>&1; echo "prof ssr rc=$?" )
python scripts/prof_steps.py $(ls $O/prof_ssr/*/*_results.db $O/prof_ssr/*_results.db 2>/dev/null | head -1) --csv $O/kernel_stats_ssr.csv > $O/kernel_stats_ssr.txt 2>&1; head -n 12 $O/kernel_stats_ssr.txt
rm -rf $O/prof_ssr
bash scripts/pmc_passes.sh $O/pmc --precision 2
python scripts/pmc_report.py $O/pmc 150 > $O/pmc_report.txt 2>&1; head -n 45 $O/pmc_report.txt
rm -rf $O/pmc/*/*.db $O/pmc/*/*/*.db $O/prof   # the databases are large; the reports stay
grep -c . $O/smi.txt; grep "GPU use (%): 100" $O/smi.txt | tail -n 3 | cut -c1-400
# the GPU suite LAST (the measurements above are what a short budget must not lose); VFX_FINAL_TESTS=0 skips it
if [ "${VFX_FINAL_TESTS:-1}" != 0 ]; then
  timeout ${VFX_FINAL_TESTS_TIMEOUT:-1500} python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -n 3 $O/gpu_tests.log
fi
ls $O
