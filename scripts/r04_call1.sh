#!/bin/bash
# Round 4, GPU call 1: the fp16 residual trunk of the vocoder's 16-bit mode (default) against VFX_TUNE_F32_TRUNK (= round 3's data
# path): kernel + model tests, per-layer A/B of the vocoder, one bench line with f32_trunk_mode inside.
O=gpurun_out/r04c1
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "resblock or wide or f32_trunk" > $O/tests_kernels.log 2>&1; tail -n 5 $O/tests_kernels.log
timeout 1200 python -m pytest tests/test_gpu_models.py -m gpu -x -q > $O/tests_models.log 2>&1; tail -n 5 $O/tests_models.log
for t in 0 64; do
  timeout 200 python scripts/voc_layers.py tuning_$t --tuning=$t --reps=5 --json=$O/ab.jsonl > $O/ab_$t.txt 2>&1
done
grep -h "==\|GEMM-shaped" $O/ab_*.txt
grep -h "d=" $O/ab_0.txt | head -40
grep -h "per kernel" -A 12 $O/ab_0.txt
grep -h "per kernel" -A 12 $O/ab_64.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-aux --cpu-baseline-clips 2 --cpu-repeats 1 > $O/bench_gsr.json 2> $O/bench_gsr.err; cut -c1-200 $O/bench_gsr.json; tail -n 2 $O/bench_gsr.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r04c1/bench_gsr.json"))
for k in ("value","ms_per_step","ms_per_step_min","ms_per_step_median","ms_per_step_p90","step_at_ref_clock","parity","step","f16_saturated"): print(k, d.get(k))
print("f32_trunk_mode", d.get("f32_trunk_mode")); print("split", d.get("split_bf16_mode"))
print("power", d.get("power"))
r=d["roofline"]; print(r["kernel"][:40], r["bound"], r["frac"], r["traffic"])
for k,v in r["all_conv_kernels"].items(): print(k, v["launches_per_step"], v["ms_per_step"], v["frac_mfma"], v["frac_hbm"], v["hbm_bytes_per_launch"])
P
ls $O
