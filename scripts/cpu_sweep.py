import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from oracle import pipeline
from voicefixer_main_amd import synth
u, v = synth.make_resunet_state_dict(0), synth.make_vocoder_state_dict(1)
wav = synth.make_clips(1, 1.0)
for n in (8, 16, 32, 64):
    torch.set_num_threads(n)
    t = time.perf_counter(); pipeline.restore_gsr(u, v, wav); print(n, "threads", round(time.perf_counter() - t, 2), "s per 1 s clip", flush=True)
