#!/bin/bash
# ONE parameterised GPU call (replaces the per-call scripts of rounds 3-4, which are in the history: `git log -- scripts/`).
#
#   scripts/measure.sh <tag> <step> [<step> ...]      ->  gpurun_out/<tag>/...
#
# Steps run in the order given; a failing step does not stop the later ones (a short GPU budget must not lose a measurement
# to an unrelated failure).  Arguments of a step follow a colon; commas stand for spaces inside them.
#
#   smoke                      __graft_entry__.smoke()
#   tests[:K]                  pytest tests -m gpu [-k K]               -> tests[_K].log
#   tfile:F[:K]                pytest tests/F -m gpu -x [-k K]           -> tfile_F.log
#   bench[:ARGS]               python bench.py --steps 20 --warmup 5 ARGS  (the judged line; no poller beside it) -> bench[_ARGS].json
#   quick[:ARGS]               bench.py without parity / roofline / aux / alt / cpu baseline (timing only)       -> quick.jsonl (appended)
#   ab:DIR[:N]                 DIR (a built tree of another commit, e.g. _ab_r04/) and this tree alternating N times (default 2),
#                              `quick` flags, on this one box                                                      -> same_box_ab.jsonl
#   lib:NAME:STEP...           the rest of the step with VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_NAME.so (scripts/build_variant.sh)
#   handler[:ARGS]             scripts/bench_handler.py ARGS                                                        -> handler.json
#   varlen[:ARGS]              scripts/bench_varlen.py ARGS (mixed-length clips through vfx_restore_gsr_varlen)    -> varlen.json
#   unet[:ARGS]                scripts/unet_time.py <tag> ARGS (the mel ResUNet alone at the benched shape)         -> unet.jsonl
#   voc[:ARGS]                 scripts/voc_layers.py ARGS (per-layer medians of the vocoder)                        -> voc_layers.txt
#   prof[:ARGS]                rocprofv3 --kernel-trace --stats of a 5-step bench run ARGS                          -> kernel_stats.{csv,txt}
#   pmc[:ARGS]                 scripts/pmc_passes.sh + pmc_report.py (one counter group per pass, kernel-trace only) -> pmc_report.txt
#   phase                      scripts/phase_timing.py on voicefixer_main_amd/abl/libvfx_timing.so (scripts/build_timing.sh) -> phase_timing.txt
#   py:SCRIPT[:ARGS]           python scripts/SCRIPT ARGS                                                           -> SCRIPT.log
tag=${1:?usage: scripts/measure.sh <tag> <step> ...}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/$tag
mkdir -p $O
QUICK="--no-aux --no-alt --cpu-baseline-clips 0 --no-parity --traffic off --no-roofline"
args() { echo "${1//,/ }"; }
slug() { echo "$1" | tr -c 'A-Za-z0-9_.\n' '_' | cut -c1-40; }
line() { python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); d['tree']='$1'; print(json.dumps(d))"; }
run_step() {
  local step=$1 name rest
  name=${step%%:*}; rest=""; [ "$name" != "$step" ] && rest=${step#*:}
  case $name in
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -n 3 $O/smoke.log ;;
    tests) local k=$(args "$rest"); local f=$O/tests$( [ -n "$k" ] && echo _$(slug "$k") ).log
           timeout ${VFX_TESTS_TIMEOUT:-1500} python -m pytest tests -m gpu -q ${k:+-k "$k"} > $f 2>&1; tail -n 4 $f ;;
    tfile) local f=${rest%%:*} k=""; [ "$f" != "$rest" ] && k=$(args "${rest#*:}")
           timeout 900 python -m pytest tests/$f -m gpu -x -q ${k:+-k "$k"} > $O/tfile_$(slug "$f").log 2>&1; tail -n 6 $O/tfile_$(slug "$f").log ;;
    bench) local a=$(args "$rest"); local f=$O/bench$( [ -n "$a" ] && echo _$(slug "$a") )
           timeout 900 python bench.py --steps 20 --warmup 5 $a > $f.json 2> $f.err; cut -c1-220 $f.json; tail -n 2 $f.err ;;
    quick) timeout 300 python bench.py --steps 20 --warmup 5 $QUICK $(args "$rest") 2>/dev/null | line "${VFX_LIB_PATH:-this tree} $rest" >> $O/quick.jsonl; tail -n 1 $O/quick.jsonl | cut -c1-200 ;;
    ab) local d=${rest%%:*} n=2; [ "$d" != "$rest" ] && n=${rest#*:}
        for i in $(seq $n); do
          ( cd $d && timeout 300 python bench.py --steps 20 --warmup 5 $QUICK 2>/dev/null | line "$d" ) >> $O/same_box_ab.jsonl
          timeout 300 python bench.py --steps 20 --warmup 5 $QUICK 2>/dev/null | line "this tree" >> $O/same_box_ab.jsonl
        done
        python -c "import json
for l in open('$O/same_box_ab.jsonl'):
    d=json.loads(l); print(d['tree'], d['value'], d['ms_per_step'], d.get('ms_per_step_median'), (d.get('power') or {}).get('avg_sclk_mhz'))" ;;
    lib) local v=${rest%%:*}; ( export VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_$v.so; run_step "${rest#*:}" ) ;;
    handler) timeout 300 python scripts/bench_handler.py $(args "$rest") > $O/handler.json 2> $O/handler.err; cut -c1-400 $O/handler.json; tail -n 2 $O/handler.err ;;
    varlen) timeout 600 python scripts/bench_varlen.py $(args "$rest") > $O/varlen.json 2> $O/varlen.err; cut -c1-600 $O/varlen.json; tail -n 2 $O/varlen.err ;;
    unet) timeout 200 python scripts/unet_time.py "${VFX_LIB_PATH:-$tag}" --reps=10 --json=$O/unet.jsonl $(args "$rest") 2>&1 | grep -v "^\s*$" | tee -a $O/unet.txt | tail -n 70 ;;
    voc) timeout 300 python scripts/voc_layers.py $(args "$rest") >> $O/voc_layers.txt 2>&1; tail -n 45 $O/voc_layers.txt ;;
    prof) ( cd /tmp; export TMPDIR=/tmp; timeout 400 rocprofv3 --kernel-trace --stats -d "$ROOT/$O/prof" -o $tag -- \
              python "$ROOT/bench.py" --steps 5 --warmup 2 --no-alt --no-aux --cpu-baseline-clips 0 --traffic off --no-parity $(args "$rest") > "$ROOT/$O/prof.log" 2>&1; echo "prof rc=$?" )
          python scripts/prof_steps.py $(ls $O/prof/*/*_results.db $O/prof/*_results.db 2>/dev/null | head -1) --csv $O/kernel_stats.csv > $O/kernel_stats.txt 2>&1
          head -n 16 $O/kernel_stats.txt; rm -rf $O/prof ;;
    pmc) bash scripts/pmc_passes.sh $O/pmc --precision 2 $(args "$rest"); python scripts/pmc_report.py $O/pmc 150 > $O/pmc_report.txt 2>&1
         head -n 45 $O/pmc_report.txt; rm -rf $O/pmc/*/*.db $O/pmc/*/*/*.db ;;
    phase) VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_timing.so timeout 300 python scripts/phase_timing.py $(args "$rest") > $O/phase_timing.txt 2>&1; tail -n 40 $O/phase_timing.txt ;;
    py) local s=${rest%%:*} a=""; [ "$s" != "$rest" ] && a=$(args "${rest#*:}")
        timeout 600 python scripts/$s $a > $O/$(slug "$s").log 2>&1; tail -n 30 $O/$(slug "$s").log ;;
    *) echo "measure.sh: unknown step '$step'" ;;
  esac
}
for step in "$@"; do
  echo "=== $step"
  run_step "$step"
done
ls $O
