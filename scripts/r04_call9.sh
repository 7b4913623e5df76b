#!/bin/bash
# Round 4, GPU call 9: the short-clip split-K rule in the product library -- full GPU suite, the three workloads it could touch.
O=gpurun_out/r04c9
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; tail -n 4 $O/gpu_tests.log
timeout 120 python scripts/unet_time.py product --reps=10 --json=$O/unet.jsonl 2>&1 | grep "=="
timeout 300 python bench.py --workload stream1s --steps 100 --warmup 10 > $O/bench_stream1s.json 2> $O/bench_stream1s.err; python -c "import json; d=json.load(open('$O/bench_stream1s.json')); print('stream1s', d['value'], d['ms_per_step'], d.get('parity'))"
timeout 300 python bench.py --workload ssr_sr64 --steps 3 --warmup 1 --no-roofline --no-parity --cpu-baseline-clips 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ssr_sr64', d['ms_per_step'], 'ms')"
ls $O
