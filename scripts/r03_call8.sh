#!/bin/bash
# Round 3, GPU call 8: resblock_r128 with the patch through registers (no LDS-DMA staging, no in-place pass) against the DMA form
# (abl/libvfx_dma64.so); bench line with the power / clock sampler.
O=gpurun_out/r03c8
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "resblock or conv" > $O/tests_kernels.log 2>&1; tail -n 3 $O/tests_kernels.log
timeout 1200 python -m pytest tests/test_gpu_models.py -m gpu -x -q > $O/tests_models.log 2>&1; tail -n 3 $O/tests_models.log
for v in dma64 default; do
  if [ $v = default ]; then unset VFX_LIB_PATH; else export VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_$v.so; fi
  timeout 200 python scripts/voc_layers.py lib_$v --reps=5 --json=$O/ab.jsonl > $O/ab_$v.txt 2>&1
done
unset VFX_LIB_PATH
grep -h "==\|k_resblock<256\|k_resblock<128" $O/ab_*.txt | grep -v "d="
grep -h "k_resblock<128.*d=" $O/ab_default.txt | head -8
VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_timing.so timeout 600 python scripts/phase_timing.py --json=$O/phase_timing.json > $O/phase_timing.txt 2>&1; grep -v amdgpu.ids $O/phase_timing.txt | grep -A13 "C = 128, d = 1:"
timeout 600 python bench.py --steps 20 --warmup 5 --no-aux --cpu-baseline-clips 0 > $O/bench_gsr.json 2> $O/bench_gsr.err; cut -c1-160 $O/bench_gsr.json; tail -n 2 $O/bench_gsr.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r03c8/bench_gsr.json"))
print("power", d.get("power")); print("step", d.get("step"))
r=d["roofline"]; print(r["kernel"], r["bound"], r["frac"], r["traffic"], (r.get("traffic_detail") or {}).get("effective_sclk_ghz_profiled"))
for k,v in r["all_conv_kernels"].items(): print(k, v["ms_per_step"], v["frac_mfma"], v["frac_hbm"])
P
ls $O
