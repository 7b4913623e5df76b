"""Why two fp32 evaluations of the spectrogram path agree to ~60 dB only on LOW-PASSED clips: decomposes the fp32-vs-float64
error of oracle.pipeline.restore_ssr into trunk / phase / input-magnitude contributions, low-passed vs full-band clips, full vs
damped residual branches (round 4).  Result: the trunk is well conditioned (120 dB); the PHASE of the numerically empty bins above
the cut-off is the limit (65-75 dB), and it disappears on full-band input (108-117 dB).  CPU only, a few minutes."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from oracle import pipeline, dsp, resunet
from voicefixer_main_amd import synth
torch.set_num_threads(8)
def sisdr(ref, est):
    e = est - ref
    return 10*np.log10((ref**2).sum()/((e**2).sum()+1e-300))
for secs, mode in ((0.6,"lowpass"),(3.0,"lowpass"),(3.0,"noise")):
    wav = synth.make_clips(1, secs, seed=7, mode=mode)
    for g in (1.0, 0.25):
        sd = synth.make_resunet_state_dict(2, res_gain=g)
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        a = pipeline.restore_ssr(sd, wav)
        b = pipeline.restore_ssr(sd64, wav, dtype=torch.float64)
        # trunk alone on the float64 spectrogram
        mag32 = resunet.unet_spec_mag(sd, torch.from_numpy(b["sp"]).float()).numpy().astype(np.float64)
        sp64, cos64, sin64 = dsp.spectrogram_phase(wav.astype(np.float64), dtype=np.float64)
        B,C,T,Fq = mag32.shape
        w_tr = dsp.istft((mag32*cos64).reshape(B*C,T,Fq), (mag32*sin64).reshape(B*C,T,Fq), wav.shape[-1], dtype=np.float64).reshape(B,C,-1)
        # phase in fp32, trunk fp64
        sp32, cos32, sin32 = dsp.spectrogram_phase(wav.astype(np.float32), dtype=np.float32)
        w_ph = dsp.istft((b["mag"]*cos32).reshape(B*C,T,Fq).astype(np.float64), (b["mag"]*sin32).reshape(B*C,T,Fq).astype(np.float64), wav.shape[-1], dtype=np.float64).reshape(B,C,-1)
        # input sp in fp32 fed to fp64 trunk
        mag_sp32 = resunet.unet_spec_mag(sd64, torch.from_numpy(sp32).double()).numpy()
        w_sp = dsp.istft((mag_sp32*cos64).reshape(B*C,T,Fq), (mag_sp32*sin64).reshape(B*C,T,Fq), wav.shape[-1], dtype=np.float64).reshape(B,C,-1)
        print("%.1fs %s gain %.2f: all-fp32 %.1f dB | trunk fp32 only %.1f | phase fp32 only %.1f | input |STFT| fp32 only %.1f" % (secs, mode, g, sisdr(b["wav"], a["wav"]), sisdr(b["wav"], w_tr), sisdr(b["wav"], w_ph), sisdr(b["wav"], w_sp)), flush=True)
