import json,sys
for f in sys.argv[1:]:
    d=json.load(open(f))
    print(f, d["value"], d["ms_per_step"], {k.split(',')[0]: v['ms_per_step'] for k,v in d["roofline"]["all_conv_kernels"].items()})
