#!/bin/bash
# Round 3, GPU call 15: the final state once more -- full GPU suite, smoke(), the default bench line (no rocm-smi poller beside it).
O=gpurun_out/r03h
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -n 3 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -n 1 $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_gsr16x10.json 2> $O/bench_gsr16x10.err; cut -c1-200 $O/bench_gsr16x10.json
ls $O
