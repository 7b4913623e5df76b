"""Degradation simulator on the host side of the boundary.

The reference builds its training / test degradations from a handful of functions; the evaluation sets the handlers are run
on (`vctk_cheby1_1000`, `..._butter_...`, evaluation_proc/config.py:91-97) are produced with them.  Mirrored here with the
same names, argument meaning and error behaviour:

* ``lowpass(data, highcut, fs, order=5, _type="butter")``            tools/dsp/lowpass.py:152-187
    ``butter`` / ``cheby1`` / ``ellip`` / ``bessel``  -> ``lowpass_filter`` (zero-phase SOS filter)    :96-133
    ``stft``       -> ``stft_hard_lowpass``: polyphase resampling down to the cut-off rate and back up      :135-146
    ``stft_hard``  -> ``stft_hard_lowpass_v0``: STFT -> zero the bins above the cut-off -> ISTFT            :21-33
  (``_type in "butter"`` is a SUBSTRING test in the reference -- ``"b"``, ``"utt"`` and ``""`` all select the Butterworth
  filter; kept, since a config that relied on it must keep working);
* ``bandpass(data, lowcut, highcut, fs, order=5, _type="butter")``     tools/dsp/lowpass.py:189-215 (IIR types, same dispatch)
* ``bandpass_filter`` / ``align_length`` / ``limit``                       tools/dsp/lowpass.py:35-94,148-150
* ``add_noise_and_scale`` / ``add_noise_and_scale_with_HQ`` / ``add_noise_and_scale_with_HQ_with_Aug``
                                                                           dataloaders/augmentation/base.py:33-118
  with their helpers ``normalize_energy`` / ``unify_energy`` (peak based: tools/others/audio_op.py:12-56).

Everything is NumPy / SciPy on the host EXCEPT ``stft_hard``: its STFT -> mask -> ISTFT round trip is the hot path's own
front-end and back-end (``vfx_stft_mel`` in its phase-emitting form and ``vfx_istft``), so it runs on the GPU through an
``Engine`` -- the reference keeps a module-level ``FDomainHelper`` for it (lowpass.py:14,111-113).
"""
import numpy as np

_engine = None     # the reference's module-level `f_helper` (lowpass.py:14): created on first use of `stft_hard`


def set_engine(engine):
    """Use this libvfx handle for `stft_hard` (otherwise one is created on cuda:0 at the first call)."""
    global _engine
    _engine = engine


def _get_engine():
    global _engine
    if _engine is None:
        from .engine import Engine
        _engine = Engine("cuda:0")
    return _engine


# ------------------------------------------------------------------------------------------------------------------
# tools/dsp/lowpass.py
# ------------------------------------------------------------------------------------------------------------------
def align_length(x, y):
    """Length of y aligned to that of x: zero-padded at the end, or cut (lowpass.py:35-55)."""
    Lx, Ly = len(x), len(y)
    if Lx == Ly:
        return y
    if Lx > Ly:
        return np.pad(y, (0, Lx - Ly), mode="constant")
    return y[:Lx]


def limit(integer, high, low):
    """lowpass.py:148-150: clamp, and truncate to int inside the range."""
    if integer > high:
        return high
    if integer < low:
        return low
    return int(integer)


def _design(order, wn, btype, ftype, what):
    from scipy.signal import bessel, butter, cheby1, cheby2, ellip
    if ftype == "butter":
        return butter(order, wn, btype=btype, output="sos")
    if ftype == "cheby1":
        return cheby1(order, 0.1, wn, btype=btype, output="sos")
    if ftype == "cheby2":
        return cheby2(order, 60, wn, btype=btype, output="sos")
    if ftype == "ellip":
        return ellip(order, 0.1, 60, wn, btype=btype, output="sos")
    if ftype == "bessel":
        return bessel(order, wn, btype=btype, output="sos")
    raise Exception("The %s filter %s is not supported!" % (what, ftype))


def lowpass_filter(x, highcut, fs, order, ftype):
    """Zero-phase (forward-backward) IIR low-pass, second-order sections (lowpass.py:96-133): 0.1 dB ripple for cheby1 /
    ellip, 60 dB stop band for cheby2 / ellip."""
    from scipy.signal import sosfiltfilt
    sos = _design(order, highcut / (0.5 * fs), "low", ftype, "lowpass")
    y = sosfiltfilt(sos, x)
    return align_length(x, y) if len(y) != len(x) else y


def bandpass_filter(x, lowcut, highcut, fs, order, ftype):
    """lowpass.py:58-94."""
    from scipy.signal import sosfiltfilt
    nyq = 0.5 * fs
    sos = _design(order, [lowcut / nyq, highcut / nyq], "band", ftype, "bandpass")
    y = sosfiltfilt(sos, x)
    return align_length(x, y) if len(y) != len(x) else y


def stft_hard_lowpass(data, lowpass_ratio, fs_ori=44100):
    """`_type="stft"` (lowpass.py:135-146): polyphase resampling down to int(ratio * fs) and back up -- the band limit of a
    recording that really was sampled at the low rate."""
    from scipy.signal import resample_poly
    fs_down = int(lowpass_ratio * fs_ori)
    y = resample_poly(data, fs_down, fs_ori)
    y = resample_poly(y, fs_ori, fs_down)
    return align_length(data, y) if len(y) != len(data) else y


def stft_hard_lowpass_v0(data, lowpass_ratio, engine=None):
    """`_type="stft_hard"` (lowpass.py:21-33): |STFT|, cos, sin of the signal (eps = 1e-8: wav_to_spectrogram_phase), the
    magnitude bins from int(1025 * ratio) up set to zero, ISTFT to the original length.  Runs on the GPU: the front-end kernel
    in its phase-emitting form (`vfx_stft_mel`), the mask, `vfx_istft`.  Returns float32 (samples,) on the host like the
    reference's `.numpy()`."""
    import torch
    eng = engine if engine is not None else _get_engine()
    length = data.shape[0]
    x = torch.as_tensor(np.ascontiguousarray(data), dtype=torch.float32)[None]
    o = eng.stft(x, want_mel=False, want_sp=True, want_phase=True)
    sp = o["sp"]
    cut = int(sp.shape[-1] * lowpass_ratio)
    sp[..., cut:] = 0.0
    return eng.istft(sp * o["cos"], sp * o["sin"], length)[0].cpu().numpy()


def lowpass(data, highcut, fs, order=5, _type="butter", engine=None):
    """lowpass.py:152-187.  data: 1-D float array (samples,) -- (samples, 1) is an error, as in the reference."""
    if len(list(data.shape)) != 1:
        raise ValueError("Error (chebyshev_lowpass_filter): Data " + str(data.shape) +
                         " should be type 1d time array, (samples,) , can not be (samples, 1)")
    # substring tests, in the reference's order
    for name in ("butter", "cheby1", "ellip", "bessel"):
        if _type in name:
            return lowpass_filter(x=data, highcut=int(highcut), fs=fs, order=limit(order, high=10, low=2), ftype=name)
    if _type in "stft":
        return stft_hard_lowpass(data, lowpass_ratio=highcut / int(fs / 2))
    if _type in "stft_hard":
        return stft_hard_lowpass_v0(data, lowpass_ratio=highcut / int(fs / 2), engine=engine)
    raise ValueError("Error: Unexpected filter type " + _type)


def bandpass(data, lowcut, highcut, fs, order=5, _type="butter"):
    """lowpass.py:189-215: the band-pass twin of `lowpass` -- same 1-D check, same substring dispatch and order clamp, IIR
    types only (butter / cheby1 / ellip / bessel; cheby2 is commented out in the reference and raises here as well)."""
    if len(list(data.shape)) != 1:
        raise ValueError("Error (chebyshev_lowpass_filter): Data " + str(data.shape) +
                         " should be type 1d time array, (samples,) , can not be (samples, 1)")
    for name in ("butter", "cheby1", "ellip", "bessel"):
        if _type in name:
            return bandpass_filter(x=data, lowcut=int(lowcut), highcut=int(highcut), fs=fs,
                                   order=limit(order, high=10, low=2), ftype=name)
    raise ValueError("Error: Unexpected filter type " + _type)


# ------------------------------------------------------------------------------------------------------------------
# tools/others/audio_op.py:12-56 (peak-based "energy") and dataloaders/augmentation/base.py:33-118
# ------------------------------------------------------------------------------------------------------------------
def activelev(*args):
    """Largest absolute sample over all the signals (audio_op.py:41-56)."""
    return max(float(np.max(np.abs(np.asarray(a)))) for a in args)


def normalize_energy(audio, alpha=1):
    """Peak to alpha (audio_op.py:12-29)."""
    return (audio / activelev(audio)) * alpha


def unify_energy(*args):
    """All signals by ONE factor so that the largest peak among them becomes 1 (audio_op.py:31-39)."""
    s = 1.0 / activelev(*args)
    return [x * s for x in args]


def _uniform(lower, upper, rng):
    """tools/pytorch/random_.py:28-31: the upper bound itself when the interval is (almost) empty."""
    if abs(lower - upper) < 1e-5:
        return upper
    return float((upper - lower) * rng.random() + lower)


def _random_noise(clean, noise, snr_l, snr_h, rng):
    """base.py:112-115: the NOISE is divided by 10 ** (snr / 20), snr ~ U[snr_l, snr_h) dB."""
    snr = _uniform(snr_l, snr_h, rng)
    return clean, noise / (10 ** (float(snr) / 20)), snr


def add_noise_and_scale(front, noise, snr_l=-5, snr_h=35, scale_lower=0.6, scale_upper=1.0, rng=None):
    """base.py:33-54: both signals to unit peak, the noise lowered by a random SNR (peak ratio, dB), the mixture's peak
    to 1 (all three by the same factor), a random common scale.  -> (front, noise, snr, scale); noisy = front + noise."""
    rng = rng if rng is not None else np.random.default_rng()
    snr = None
    noise, front = normalize_energy(noise), normalize_energy(front)
    if snr_l is not None and snr_h is not None:
        front, noise, snr = _random_noise(front, noise, snr_l, snr_h, rng)
    _, noise, front = unify_energy(noise + front, noise, front)
    scale = _uniform(scale_lower, scale_upper, rng)
    return front * scale, noise * scale, snr, scale


def _match_noise_level(noise, level_of):
    """base.py:74-78 / :101-105: "some clipping noise is extremely noisy" -- unless the speech is nearly silent, the noise's
    mean absolute level is set to the speech's before the SNR is applied."""
    front_level = float(np.mean(np.abs(level_of)))
    if front_level > 0.02:
        noise = noise / (float(np.mean(np.abs(noise))) / front_level)
    return noise


def add_noise_and_scale_with_HQ(HQ, front, noise, snr_l=-5, snr_h=35, scale_lower=0.6, scale_upper=1.0, rng=None):
    """base.py:86-110 -> (HQ, front, noise, snr, scale)."""
    rng = rng if rng is not None else np.random.default_rng()
    snr = None
    noise = normalize_energy(noise)
    HQ, front = unify_energy(HQ, front)
    noise = _match_noise_level(noise, front)
    if snr_l is not None and snr_h is not None:
        front, noise, snr = _random_noise(front, noise, snr_l, snr_h, rng)
    _, noise, front, HQ = unify_energy(noise + front, noise, front, HQ)
    scale = _uniform(scale_lower, scale_upper, rng)
    return HQ * scale, front * scale, noise * scale, snr, scale


def add_noise_and_scale_with_HQ_with_Aug(HQ, front, augfront, noise, snr_l=-5, snr_h=35, scale_lower=0.6, scale_upper=1.0,
                                         rng=None):
    """base.py:56-84 -> (HQ, front, augfront, noise, snr, scale); the noise is mixed into the AUGMENTED speech."""
    rng = rng if rng is not None else np.random.default_rng()
    snr = None
    noise = normalize_energy(noise)
    HQ, front, augfront = unify_energy(HQ, front, augfront)
    noise = _match_noise_level(noise, augfront)
    if snr_l is not None and snr_h is not None:
        augfront, noise, snr = _random_noise(augfront, noise, snr_l, snr_h, rng)
    _, augfront, noise, front, HQ = unify_energy(noise + augfront, augfront, noise, front, HQ)
    scale = _uniform(scale_lower, scale_upper, rng)
    return HQ * scale, front * scale, augfront * scale, noise * scale, snr, scale


def hard_clip(x, threshold):
    """The declipping test sets' degradation: samples limited to +-threshold (config/vctk_base_voicefixer_unet.json:80-100)."""
    return np.clip(x, -threshold, threshold)
