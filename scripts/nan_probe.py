"""Debug aid: which (precision, B, T) make the vocoder / restore read workspace nobody wrote
(run with VFX_POISON_ARENA=2 VFX_DEBUG_NAN=1)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicefixer_main_amd import synth
from voicefixer_main_amd.engine import Engine, MODEL_UNET_MEL, MODEL_VOCODER
rng = np.random.default_rng(3)
for prec in [int(a) for a in sys.argv[1:]] or [1]:
    eng = Engine("cuda:0", config={"precision": prec})
    eng.load_state_dict(MODEL_UNET_MEL, synth.make_resunet_state_dict(0))
    eng.load_state_dict(MODEL_VOCODER, synth.make_vocoder_state_dict(1))
    for B, T in [(1, 21), (2, 10), (2, 40), (2, 101)]:
        mel = (10.0 ** (rng.normal(size=(B, T, 128)) * 1.2 - 2.5)).astype(np.float32)
        print("---- vocoder prec", prec, "B", B, "T", T, flush=True)
        y = eng.vocoder(torch.from_numpy(mel))
        n = int((~torch.isfinite(y)).sum())
        print("nonfinite", n, "of", y.numel(), flush=True)
    for sec in (0.8, 1.0):
        wav = torch.from_numpy(synth.make_clips(2, sec, seed=7)[:, 0])
        print("---- restore prec", prec, "sec", sec, flush=True)
        y = eng.restore_gsr(wav)
        print("nonfinite", int((~torch.isfinite(y)).sum()), "of", y.numel(), flush=True)
