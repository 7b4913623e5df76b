#!/bin/bash
# Round 3, GPU call 11: the ResUNet's ConvTranspose2d as two phased launches (column parities = cout halves of a (B, 2H, W, 2C)
# view) instead of four (abl/libvfx_ring3.so = the four-launch form).
O=gpurun_out/r03c11
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_surface.py tests/test_gpu_shapes.py -m gpu -x -q > $O/tests_models.log 2>&1; tail -n 3 $O/tests_models.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "transpose or unet or conv" > $O/tests_kernels.log 2>&1; tail -n 3 $O/tests_kernels.log
for v in ring3 default; do
  if [ $v = default ]; then unset VFX_LIB_PATH; else export VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_$v.so; fi
  timeout 600 python bench.py --steps 10 --warmup 3 --cpu-baseline-clips 0 --traffic off --no-alt --aux-steps 5 > $O/bench_$v.json 2> $O/bench_$v.err; cut -c1-160 $O/bench_$v.json
done
unset VFX_LIB_PATH
python - <<'P'
import json
for v in ("ring3","default"):
    d=json.load(open("gpurun_out/r03c11/bench_%s.json"%v))
    r=d["roofline"]["all_conv_kernels"]
    print(v, d["value"], d["ms_per_step"], d.get("parity"), {k:(r[k]["launches_per_step"], r[k]["ms_per_step"]) for k in r if "f16" not in k})
    for k,a in d["aux_workloads"].items(): print("   ", k, a.get("value"), a.get("ms_per_step"), (a.get("parity") or {}).get("wav_sisdr_db"))
P
ls $O
