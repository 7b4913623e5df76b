for pad in 0 8192 32768 65536; do
  VFX_RB_LDS_PAD=$pad timeout 200 python bench.py --steps 5 --warmup 2 --traffic off --no-alt --cpu-baseline-clips 0 > gpurun_out/occ_$pad.json 2>/dev/null
  python - <<P
import json
d=json.load(open('gpurun_out/occ_$pad.json')); r=d['roofline']['all_conv_kernels']
print('pad $pad', d['ms_per_step'], {k:v['ms_per_step'] for k,v in r.items() if 'resblock' in k})
P
done
