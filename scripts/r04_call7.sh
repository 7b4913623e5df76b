#!/bin/bash
# Round 4, GPU call 7: the default bench line again (no rocm-smi poller beside it), with the per-launch table kept.
O=gpurun_out/r04m2
mkdir -p $O
VFX_PROFILE_DUMP=$O/convs_per_launch.csv timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_gsr16x10.json 2> $O/bench_gsr16x10.err; cut -c1-200 $O/bench_gsr16x10.json; tail -n 2 $O/bench_gsr16x10.err
python - <<'P'
import json, csv, collections
d=json.load(open("gpurun_out/r04m2/bench_gsr16x10.json"))
for k in ("value","ms_per_step","ms_per_step_median","parity","f16_saturated"): print(k, d.get(k))
r=d["roofline"]; print(r["kernel"][:40], r["bound"], r["frac"], r["avg_launch_us"], r["traffic"], r["all_conv_ms_per_step"])
for k,v in r["all_conv_kernels"].items(): print(k, v["launches_per_step"], v["ms_per_step"], v["frac_mfma"], v["frac_hbm"])
rows=list(csv.DictReader(open("gpurun_out/r04m2/convs_per_launch.csv")))
per=collections.defaultdict(list)
for x in rows:
    if x["kernel"].startswith("k_resblock<256"): per[x["Wi"]].append(float(x["ms"]))
for dil,v in sorted(per.items(), key=lambda kv:int(kv[0])): v=sorted(v); print("d=%s n=%d min %.3f med %.3f max %.3f"%(dil,len(v),v[0],v[len(v)//2],v[-1]))
P
