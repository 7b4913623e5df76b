#!/bin/bash
# Round 4, GPU call 5: the fix of the saturation record (mask before record), k_conv's 256-cout tile (64-cout waves) against
# VFX_TUNE_NO_WIDE_CONV, kernel + model tests.
O=gpurun_out/r04c5
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "resblock or wide or f32_trunk or conv" > $O/tests_kernels.log 2>&1; tail -n 4 $O/tests_kernels.log
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_shapes.py tests/test_simulate.py -m gpu -x -q -k "not ssr and not 60s" > $O/tests_models.log 2>&1; tail -n 4 $O/tests_models.log
for t in 128 0; do
  timeout 200 python scripts/voc_layers.py tuning_$t --tuning=$t --reps=5 --json=$O/ab.jsonl > $O/ab_$t.txt 2>&1
done
grep -h "==\|GEMM-shaped" $O/ab_*.txt
for t in 128 0; do echo "-- tuning $t"; grep -h "per kernel" -A 10 $O/ab_$t.txt | tail -n 10; done
VFX_PROFILE_DUMP=$O/convs_per_launch.csv timeout 600 python bench.py --steps 20 --warmup 5 --no-aux --cpu-baseline-clips 2 --cpu-repeats 1 > $O/bench_gsr.json 2> $O/bench_gsr.err; cut -c1-160 $O/bench_gsr.json; tail -n 2 $O/bench_gsr.err
python - <<'P'
import json, csv
d=json.load(open("gpurun_out/r04c5/bench_gsr.json"))
for k in ("value","ms_per_step","ms_per_step_min","ms_per_step_median","ms_per_step_p90","parity","f16_saturated","parity_failed"): print(k, d.get(k))
print("f32_trunk_mode", {k:v for k,v in d.get("f32_trunk_mode",{}).items() if k!='parity'})
r=d["roofline"]; print(r["kernel"][:40], r["bound"], r["frac"], r["traffic"])
for k,v in r["all_conv_kernels"].items(): print(k, v["launches_per_step"], v["ms_per_step"], v["frac_mfma"], v["frac_hbm"], v["hbm_bytes_per_launch"])
rows=list(csv.DictReader(open("gpurun_out/r04c5/convs_per_launch.csv")))
for r in rows[-len(rows)//20:]:
    if r['kernel'].startswith('k_conv') and 'f16' in r['kernel']: print(r['kernel'], r['M'], r['Cout'], r['K'], r['ms'], r['tflops'])
P
ls $O
