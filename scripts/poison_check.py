#!/usr/bin/env python3
"""Run every stage once on a NaN-poisoned arena (VFX_POISON_ARENA=1) and report non-finite outputs."""
import os
import sys
os.environ["VFX_POISON_ARENA"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from voicefixer_main_amd import synth
from voicefixer_main_amd.engine import Engine, MODEL_UNET_MEL, MODEL_UNET_SPEC, MODEL_VOCODER

for prec in (1, 0):
    eng = Engine("cuda:0", config={"precision": prec})
    eng.load_state_dict(MODEL_UNET_MEL, synth.make_resunet_state_dict(0))
    eng.load_state_dict(MODEL_VOCODER, synth.make_vocoder_state_dict(1))
    eng.load_state_dict(MODEL_UNET_SPEC, synth.make_resunet_state_dict(2))
    wav = torch.from_numpy(synth.make_clips(2, 1.0, seed=41)[:, 0]).cuda()
    for name, fn in (
        ("resunet_spec 1x1s", lambda: eng.resunet_spec(eng.stft(wav[:1].contiguous(), want_mel=False, want_sp=True)["sp"], wav[:1].contiguous())),
        ("resunet_mel 2x1s", lambda: eng.resunet_mel(eng.stft(wav)["mel"])),
        ("vocoder 2x1s", lambda: eng.vocoder(eng.stft(wav)["mel"])),
        ("restore_gsr 2x1s", lambda: eng.restore_gsr(wav)),
        ("resunet_spec 2x1s", lambda: eng.resunet_spec(eng.stft(wav, want_mel=False, want_sp=True)["sp"], wav)),
    ):
        y = fn()
        torch.cuda.synchronize()
        bad = int((~torch.isfinite(y)).sum().item())
        print("precision %d  %-20s non-finite: %d / %d" % (prec, name, bad, y.numel()))
