#!/bin/bash
# Round 3, GPU call 13: the fused 2-D ConvBlockRes kernels (k_resblock<32,2>, <64,4>) with the patch through registers instead of
# LDS-DMA + in-place read-back (abl/libvfx_dma2d.so = the DMA form).
O=gpurun_out/r03c13
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q > $O/tests_kernels.log 2>&1; tail -n 3 $O/tests_kernels.log
timeout 1500 python -m pytest tests/test_gpu_surface.py tests/test_gpu_shapes.py -m gpu -x -q > $O/tests_models.log 2>&1; tail -n 3 $O/tests_models.log
for v in dma2d default dma2d default; do
  if [ $v = default ]; then unset VFX_LIB_PATH; else export VFX_LIB_PATH=voicefixer_main_amd/abl/libvfx_$v.so; fi
  timeout 600 python bench.py --steps 10 --warmup 3 --cpu-baseline-clips 0 --traffic off --no-alt --aux-steps 3 > $O/bench_$v.json 2> $O/bench_$v.err
  python - <<P
import json
d=json.load(open("$O/bench_$v.json"))
r=d["roofline"]["all_conv_kernels"]
print("$v", d["value"], d["ms_per_step"], {k:r[k]["ms_per_step"] for k in r if "f16" not in k}, {k:(a.get("ms_per_step"), (a.get("parity") or {}).get("wav_sisdr_db")) for k,a in d["aux_workloads"].items()})
P
done
unset VFX_LIB_PATH
ls $O
