#!/bin/bash
# Round 4, GPU call 4: C = 512 stack on the activated-only trunk (k_conv with an activated residual), h-write VALU diet, wide
# folded tiles at C = 64 -- kernel + model tests, then the same-box A/B against the previous commit's library.
O=gpurun_out/r04c4
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "resblock or wide or f32_trunk or conv" > $O/tests_kernels.log 2>&1; tail -n 4 $O/tests_kernels.log
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_shapes.py -m gpu -x -q -k "not ssr and not 60s" > $O/tests_models.log 2>&1; tail -n 4 $O/tests_models.log
for v in prev default; do
  if [ $v = default ]; then unset VFX_LIB_PATH; else export VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_$v.so; fi
  timeout 200 python scripts/voc_layers.py lib_$v --reps=5 --json=$O/ab.jsonl > $O/ab_$v.txt 2>&1
done
unset VFX_LIB_PATH
grep -h "==\|GEMM-shaped" $O/ab_*.txt
for v in prev default; do echo "-- $v"; grep -h "per kernel" -A 9 $O/ab_$v.txt | tail -n 9; grep -h "d=" $O/ab_$v.txt | grep "<64, 8>\|d=1 \|d=243"; done
timeout 600 python bench.py --steps 20 --warmup 5 --no-aux --cpu-baseline-clips 2 --cpu-repeats 1 > $O/bench_gsr.json 2> $O/bench_gsr.err; cut -c1-160 $O/bench_gsr.json; tail -n 2 $O/bench_gsr.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r04c4/bench_gsr.json"))
for k in ("value","ms_per_step","ms_per_step_min","ms_per_step_median","ms_per_step_p90","parity","f16_saturated"): print(k, d.get(k))
print("f32_trunk_mode", {k:v for k,v in d.get("f32_trunk_mode",{}).items() if k!='parity'}, d.get("f32_trunk_mode",{}).get("parity",{}).get("wav_sisdr_db"))
r=d["roofline"]; print(r["kernel"][:40], r["bound"], r["frac"], r["traffic"])
for k,v in r["all_conv_kernels"].items(): print(k, v["launches_per_step"], v["ms_per_step"], v["frac_mfma"], v["frac_hbm"], v["hbm_bytes_per_launch"])
P
ls $O
