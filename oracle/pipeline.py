"""Oracle glue + whole-path restatements (numpy / torch CPU).

Test infrastructure only (see oracle/__init__.py).

Restates:
* ``trim_center``        tools/utils.py:57-70
* ``amp_to_original_f``  tools/utils.py:50-55
* the per-segment body of ``handler`` eval_gsr_voicefixer.py:47-74
  (pre -> model -> from_log -> [energy unify] -> vocoder -> peak normalise -> trim_center)
* ``UNetResComplex_100Mb.forward`` (spectrogram) + ``Generator.forward`` of ssr_unet:
  models/components/unet_v2.py:86-148, models/ssr_unet.py:51-54
"""
import numpy as np
import torch

from . import dsp, resunet
from . import vocoder as voc


def trim_center(est, ref):
    """Centre-crop the longer of (est, ref) along the last axis; tools/utils.py:57-70.

    Faithful to the reference including its quirk: the slice ``[d//2 : -(d//2)]``
    is empty when the length difference is 1 (``-0`` == 0).
    """
    diff = abs(est.shape[-1] - ref.shape[-1])
    if diff == 0:
        return est, ref
    min_len = min(est.shape[-1], ref.shape[-1])
    h = int(diff // 2)
    if est.shape[-1] > ref.shape[-1]:
        est = est[..., h:-h]
    else:
        ref = ref[..., h:-h]
    return est[..., :min_len], ref[..., :min_len]


def amp_to_original_f(mel_est, mel_target, cutoff=0.2):
    """Match the mean energy of mel bins 5..int(128*0.2) of the estimate to the target."""
    hi = int(mel_target.shape[-1] * cutoff)
    e_est = mel_est[..., 5:hi].mean(axis=(2, 3))
    e_tgt = mel_target[..., 5:hi].mean(axis=(2, 3))
    return mel_est * (e_tgt / e_est)[..., None, None], mel_target


def peak_normalise(wav):
    """eval_gsr_voicefixer.py:68-70: divide by max|x| iff it exceeds 1 (per call, whole tensor)."""
    peak = np.abs(wav).max()
    return wav / peak if peak > 1.0 else wav


def restore_gsr(unet_sd, voc_sd, wav, unify_energy=False, dtype=torch.float32, cfg=voc.VocoderConfig()):
    """One `handler` segment for a batch of clips: wav (B,1,L) float -> dict of stage outputs.

    The reference evaluates batch 1; the peak normalisation is therefore applied per
    clip here (each clip is its own `handler` call).
    """
    npd = np.float32 if dtype == torch.float32 else np.float64
    sp, mel = dsp.wav_to_mel(np.asarray(wav, npd), dtype=npd)
    logmel = resunet.generator_mel(unet_sd, torch.from_numpy(mel).to(dtype)).numpy()
    den = dsp.from_log(logmel)
    if unify_energy:
        den, _ = amp_to_original_f(den, mel)
    out = voc.vocoder(voc_sd, torch.from_numpy(den).to(dtype), cfg).numpy()
    out = np.stack([peak_normalise(o) for o in out])
    out, _ = trim_center(out, np.asarray(wav))
    return {"mel_in": mel, "logmel": logmel, "mel_out": den, "wav": out}


def restore_ssr(unet_sd, wav, dtype=torch.float32):
    """ssr_unet / gsr_unet forward: wav (B,1,L) -> {'wav': (B,1,L)}.

    sp = |STFT(wav)| is the network input (linear magnitude, eval_ssr_unet.py:112-113);
    the trunk's output magnitude is recombined with the INPUT phase and inverted.
    """
    npd = np.float32 if dtype == torch.float32 else np.float64
    wav = np.asarray(wav, npd)
    sp, cos, sin = dsp.spectrogram_phase(wav, dtype=npd)
    mag = resunet.unet_spec_mag(unet_sd, torch.from_numpy(sp).to(dtype)).numpy()
    B, C, T, Fq = mag.shape
    out = dsp.istft((mag * cos).reshape(B * C, T, Fq), (mag * sin).reshape(B * C, T, Fq), wav.shape[-1], dtype=npd)
    return {"sp": sp, "mag": mag, "wav": out.reshape(B, C, -1)}
