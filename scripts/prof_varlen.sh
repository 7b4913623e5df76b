#!/bin/bash
# rocprofv3 kernel trace of the shipped mixed-length path alone (scripts/bench_varlen.py --only-varlen): per-kernel totals
#   bash scripts/prof_varlen.sh <tag> [clips]      ->  gpurun_out/<tag>/varlen_kernel_stats.{csv,txt}
tag=${1:-varlen}; n=${2:-128}
O=gpurun_out/$tag; mkdir -p $O
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
( cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats -d "$ROOT/$O/prof" -o vl -- python "$ROOT/scripts/bench_varlen.py" --clips=$n --reps=2 --only-varlen > "$ROOT/$O/prof.log" 2>&1; echo "prof rc=$?" )
python scripts/prof_steps.py $(ls $O/prof/*/*_results.db $O/prof/*_results.db 2>/dev/null | head -1) --csv $O/varlen_kernel_stats.csv > $O/varlen_kernel_stats.txt 2>&1
head -n 30 $O/varlen_kernel_stats.csv; rm -rf $O/prof
