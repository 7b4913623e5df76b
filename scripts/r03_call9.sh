#!/bin/bash
# Round 3, GPU call 9: the single-form C = 256 layer (resblock_s256.hip, 64-position tiles, x read once) against the two-form
# layer of resblock_w64.hip (vfx_config.tuning = 256).
O=gpurun_out/r03c9
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "resblock or wide" > $O/tests_kernels.log 2>&1; tail -n 3 $O/tests_kernels.log
timeout 1200 python -m pytest tests/test_gpu_models.py -m gpu -x -q > $O/tests_models.log 2>&1; tail -n 3 $O/tests_models.log
for t in 0 256; do
  timeout 200 python scripts/voc_layers.py tuning_$t --tuning=$t --reps=5 --json=$O/ab.jsonl > $O/ab_tuning_$t.txt 2>&1
done
grep -h "==\|k_resblock<256\|k_resblock<128\|k_conv<128" $O/ab_tuning_*.txt | grep -v "d="
grep -h "k_resblock<256.*d=" $O/ab_tuning_0.txt | head -8
VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_timing.so timeout 600 python scripts/phase_timing.py --json=$O/phase_timing.json > $O/phase_timing.txt 2>&1; grep -v amdgpu.ids $O/phase_timing.txt | grep -A13 "C = 256, d = 1:"
timeout 600 python bench.py --steps 20 --warmup 5 --no-aux --cpu-baseline-clips 0 > $O/bench_gsr.json 2> $O/bench_gsr.err; cut -c1-160 $O/bench_gsr.json; tail -n 2 $O/bench_gsr.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r03c9/bench_gsr.json"))
print("step", d.get("step")); print("parity", d.get("parity"))
r=d["roofline"]; print(r["kernel"], r["bound"], r["frac"], r["traffic"])
for k,v in r["all_conv_kernels"].items(): print(k, v["ms_per_step"], v["frac_mfma"], v["frac_hbm"], v["hbm_bytes_per_launch"])
P
ls $O
