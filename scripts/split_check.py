"""Parity of the split-bf16 operand mode against the CPU oracle (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from oracle import pipeline, resunet
from oracle import vocoder as ovoc
from voicefixer_main_amd import synth
from voicefixer_main_amd.engine import Engine, MODEL_UNET_MEL, MODEL_VOCODER
u, v = synth.make_resunet_state_dict(0), synth.make_vocoder_state_dict(1)
for prec in (0, 1):
    eng = Engine("cuda:0", config={"precision": prec})
    eng.load_state_dict(MODEL_UNET_MEL, u); eng.load_state_dict(MODEL_VOCODER, v)
    rng = np.random.default_rng(0)
    mel = (10.0 ** (rng.normal(size=(2, 1, 200, 128)) * 1.2 - 2.5)).astype(np.float32)
    ref = resunet.generator_mel({k: t.double() for k, t in u.items()}, torch.from_numpy(mel).double()).numpy()[:, 0]
    got = eng.resunet_mel(torch.from_numpy(mel[:, 0])).cpu().numpy()
    d = np.abs(got - ref)
    print("prec", prec, "unet logmel L1 %.3g max %.3g" % (d.mean(), d.max()))
    m2 = (10.0 ** (rng.normal(size=(1, 1, 40, 128)) * 1.2 - 2.5)).astype(np.float32)
    refw = ovoc.vocoder({k: t.double() for k, t in v.items()}, torch.from_numpy(m2).double()).numpy()[:, 0]
    gw = eng.vocoder(torch.from_numpy(m2[:, 0])).cpu().numpy()
    e = gw - refw
    print("prec", prec, "vocoder max err %.3g  SI-SDR %.1f dB  (peak %.3g)" % (np.abs(e).max(), 10*np.log10((refw**2).sum()/(e**2).sum()), np.abs(refw).max()))
    wav = synth.make_clips(2, 1.0)
    r = pipeline.restore_gsr(u, v, wav)
    o, lg = eng.restore_gsr(torch.from_numpy(wav[:, 0]), want_logmel=True)
    e = o.cpu().numpy() - r["wav"][:, 0]
    print("prec", prec, "restore logmel L1 %.3g  wav SI-SDR %.1f dB" % (np.abs(lg.cpu().numpy() - r["logmel"][:, 0]).mean(), 10*np.log10((r["wav"]**2).sum()/(e**2).sum())))
    eng.close()
