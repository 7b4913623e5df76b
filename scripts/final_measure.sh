#!/bin/bash
# One GPU call: GPU tests, the default bench line, a kernel-trace of the same command and the PMC passes.
#   scripts/final_measure.sh <tag>      ->  gpurun_out/{gpu_tests.log, bench_<tag>.json, prof_<tag>/, pmc_<tag>/}
tag=${1:-v7}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; tail -3 gpurun_out/gpu_tests.log
python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err; cut -c1-300 gpurun_out/bench_$tag.json
( cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/prof_$tag" -o $tag -- \
    python "$ROOT/bench.py" --steps 3 --warmup 1 --no-alt --cpu-baseline-clips 0 > "$ROOT/gpurun_out/prof_$tag.log" 2>&1; echo "prof rc=$?" )
bash scripts/pmc_passes.sh gpurun_out/pmc_$tag --precision 2
ls gpurun_out/prof_$tag gpurun_out/pmc_$tag | head -20
