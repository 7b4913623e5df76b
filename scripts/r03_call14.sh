#!/bin/bash
# Round 3, GPU call 14: the (1, 3) layer pair at C = 128 (resblock_r128.hip, PAIR) against one launch per layer
# (abl/libvfx_dma2d.so = the commit before).
O=gpurun_out/r03c14
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "resblock or wide" > $O/tests_kernels.log 2>&1; tail -n 3 $O/tests_kernels.log
timeout 1200 python -m pytest tests/test_gpu_models.py -m gpu -x -q > $O/tests_models.log 2>&1; tail -n 3 $O/tests_models.log
for v in dma2d default; do
  if [ $v = default ]; then unset VFX_LIB_PATH; else export VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_$v.so; fi
  timeout 200 python scripts/voc_layers.py lib_$v --reps=5 --json=$O/ab.jsonl > $O/ab_$v.txt 2>&1
done
unset VFX_LIB_PATH
grep -h "==\|k_resblock<256\|k_resblock<128\|k_resblock_pair<128\|GEMM-shaped" $O/ab_*.txt | grep -v "d="
grep -h "k_resblock<128.*d=\|k_resblock_pair<128.*d=" $O/ab_default.txt | head -8
timeout 600 python bench.py --steps 20 --warmup 5 --no-aux --cpu-baseline-clips 2 --cpu-repeats 1 > $O/bench_gsr.json 2> $O/bench_gsr.err; cut -c1-160 $O/bench_gsr.json; tail -n 2 $O/bench_gsr.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r03c14/bench_gsr.json"))
print("parity", d.get("parity")); print("step", d.get("step"))
r=d["roofline"]; print(r["kernel"][:40], r["bound"], r["frac"], r["traffic"])
for k,v in r["all_conv_kernels"].items(): print(k, v["launches_per_step"], v["ms_per_step"], v["frac_mfma"], v["frac_hbm"], v["hbm_bytes_per_launch"])
P
ls $O
