"""Oracle DSP: STFT / ISTFT / mel filterbank / log maps (numpy float64 or float32).

Test infrastructure only (see oracle/__init__.py).

Restates, with citations into /root/reference:

* torchlibrosa 0.0.7 ``STFT`` / ``ISTFT`` as parameterised by
  ``tools/pytorch/modules/fDomainHelper.py:26-32`` (n_fft 2048, hop 441, hann,
  center=True, reflect padding) -- third-party, un-vendored: the published
  algorithm is a windowed DFT done as two conv1d's (real / imag, imag carries
  the minus sign) and an inverse done as IDFT * window, overlap-add, division
  by the window sum-of-squares envelope and removal of the n_fft//2 centre pad.
* ``FDomainHelper.spectrogram_phase`` ``fDomainHelper.py:60-65`` (eps clamp on
  the POWER, mag = sqrt, cos = re/mag, sin = im/mag) and
  ``wav_to_spectrogram_phase`` ``:67-89`` (eps = 1e-8).
* ``melscale_fbanks`` / ``MelScale.forward`` ``tools/pytorch/mel_scale.py:156-221``,
  ``:52-64`` (HTK, f_min 0, f_max sr//2, norm None, all_freqs = linspace(0, sr//2, n)).
* ``to_log`` / ``from_log`` ``tools/pytorch/pytorch_util.py:157-163``.
"""
import math

import numpy as np

from . import HOP, N_BINS, N_FFT, N_MELS, SAMPLE_RATE


# ----------------------------------------------------------------------------
# windows and framing
# ----------------------------------------------------------------------------
def hann_periodic(n=N_FFT, dtype=np.float64):
    """librosa.filters.get_window('hann', n, fftbins=True): periodic Hann."""
    k = np.arange(n, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)).astype(dtype)


def num_frames(length, hop=HOP):
    """center=True framing: T = L // hop + 1."""
    return length // hop + 1


def reflect_pad(x, pad):
    """np.pad(..., mode='reflect') along the last axis (edge sample not repeated)."""
    return np.pad(x, [(0, 0)] * (x.ndim - 1) + [(pad, pad)], mode="reflect")


def frame_signal(x, n_fft=N_FFT, hop=HOP):
    """(B, L) -> (B, T, n_fft) frames of the reflect-padded signal."""
    xp = reflect_pad(x, n_fft // 2)
    T = num_frames(x.shape[-1], hop)
    idx = np.arange(T)[:, None] * hop + np.arange(n_fft)[None, :]
    return xp[..., idx]


# ----------------------------------------------------------------------------
# STFT (a1)
# ----------------------------------------------------------------------------
def stft(x, n_fft=N_FFT, hop=HOP, dtype=np.float64):
    """Windowed DFT of every frame.  x: (B, L) -> (re, im) each (B, T, n_fft//2+1).

    im follows the forward-DFT sign convention (im = -sum x w sin), as the
    conv_imag weights of torchlibrosa do.  `dtype` is the working precision of
    the transform (float64 = ground truth; float32 mimics an fp32 FFT).
    """
    frames = frame_signal(np.asarray(x, dtype=dtype), n_fft, hop) * hann_periodic(n_fft, dtype)
    spec = np.fft.rfft(frames.astype(np.float64), axis=-1)
    return spec.real.astype(dtype), spec.imag.astype(dtype)


def stft_dft_matrix(x, frame_ids, n_fft=N_FFT, hop=HOP):
    """Explicit float64 DFT-matrix evaluation of selected frames (cross-check for stft())."""
    frames = frame_signal(np.asarray(x, np.float64), n_fft, hop)[..., frame_ids, :] * hann_periodic(n_fft)
    n = np.arange(n_fft)[:, None]
    k = np.arange(n_fft // 2 + 1)[None, :]
    ang = 2.0 * np.pi * ((n * k) % n_fft) / n_fft
    return frames @ np.cos(ang), -(frames @ np.sin(ang))


def spectrogram_phase(x, eps=1e-8, dtype=np.float64):
    """wav (B, C, L) -> (mag, cos, sin) each (B, C, T, 1025)  [fDomainHelper.py:60-89]."""
    x = np.asarray(x)
    B, C, L = x.shape
    re, im = stft(x.reshape(B * C, L), dtype=dtype)
    power = np.maximum(re * re + im * im, dtype(eps))
    mag = np.sqrt(power)
    shp = (B, C) + re.shape[1:]
    return mag.reshape(shp), (re / mag).reshape(shp), (im / mag).reshape(shp)


# ----------------------------------------------------------------------------
# ISTFT (a10)
# ----------------------------------------------------------------------------
def window_sumsquare(n_frames, n_fft=N_FFT, hop=HOP):
    """librosa.filters.window_sumsquare(hann, n_frames, norm=None): OLA of window**2."""
    w2 = hann_periodic(n_fft) ** 2
    env = np.zeros(n_fft + hop * (n_frames - 1), np.float64)
    for t in range(n_frames):
        env[t * hop:t * hop + n_fft] += w2
    return env


def istft(re, im, length, n_fft=N_FFT, hop=HOP, dtype=np.float64):
    """(B, T, 1025) real / imag -> (B, length) waveform.

    Hermitian-extend, IDFT, multiply by the synthesis window, overlap-add, divide
    by the window sum-of-squares envelope where it is non-tiny, strip the leading
    n_fft//2 samples and keep `length` samples (`y[:, n_fft//2 : n_fft//2 + length]`).
    """
    re = np.asarray(re, np.float64)
    im = np.asarray(im, np.float64)
    B, T, _ = re.shape
    frames = np.fft.irfft(re + 1j * im, n=n_fft, axis=-1) * hann_periodic(n_fft)
    y = np.zeros((B, n_fft + hop * (T - 1)), np.float64)
    for t in range(T):
        y[:, t * hop:t * hop + n_fft] += frames[:, t]
    env = window_sumsquare(T, n_fft, hop)
    nz = env > np.finfo(np.float32).tiny
    y[:, nz] /= env[nz]
    # torchlibrosa: y[:, n_fft//2 : n_fft//2 + length]; the in-repo twin tools/dsp/base.py:193-200 does the same
    # (start = n_fft//2, end = start + length): the samples past hop*(T-1) are reconstructed from the tail of the
    # last frames, NOT zeroed.  Zeros only past the end of the overlap-add buffer (T not matching length).
    y = y[:, n_fft // 2:n_fft // 2 + length]
    out = np.zeros((B, length), np.float64)
    out[:, :y.shape[1]] = y
    return out.astype(dtype)


# ----------------------------------------------------------------------------
# mel filterbank (a2)
# ----------------------------------------------------------------------------
def _hz_to_mel_htk(f):
    return 2595.0 * math.log10(1.0 + f / 700.0)


def mel_filterbank(n_freqs=N_BINS, n_mels=N_MELS, sample_rate=SAMPLE_RATE, dtype=np.float32):
    """HTK triangular filterbank, (n_freqs, n_mels) [mel_scale.py:131-221].

    The reference builds the table with float32 torch ops (torch.linspace, pow, ...);
    its last bits depend on that arithmetic, so the oracle evaluates the same
    formula with the same float32 torch-CPU primitives and is bit-identical to the
    reference module (pinned in tests/golden).  Triangles: rising slope
    (f - f_lo)/(f_c - f_lo), falling slope (f_hi - f)/(f_hi - f_c), clipped at 0.
    """
    import torch
    bin_hz = torch.linspace(0, sample_rate // 2, n_freqs)
    mel_hi = _hz_to_mel_htk(float(sample_rate // 2))
    mel_edges = torch.linspace(_hz_to_mel_htk(0.0), mel_hi, n_mels + 2)
    edge_hz = 700.0 * (10.0 ** (mel_edges / 2595.0) - 1.0)
    width = edge_hz[1:] - edge_hz[:-1]
    dist = edge_hz[None, :] - bin_hz[:, None]                  # (n_freqs, n_mels + 2)
    rising = (-1.0 * dist[:, :-2]) / width[:-1]
    falling = dist[:, 2:] / width[1:]
    fb = torch.clamp(torch.minimum(rising, falling), min=0.0)
    return fb.numpy().astype(dtype)


def mel_project(sp, fb=None):
    """(..., T, 1025) magnitudes -> (..., T, 128); the permutes of eval_gsr_voicefixer.py:23 cancel."""
    if fb is None:
        fb = mel_filterbank(dtype=sp.dtype)
    return sp @ fb.astype(sp.dtype)


# ----------------------------------------------------------------------------
# log maps (a3)
# ----------------------------------------------------------------------------
def to_log(x):
    """log10(clip(x, 1e-8)); asserts non-negativity like pytorch_util.py:157-159."""
    x = np.asarray(x)
    assert not (x < 0).any(), "to_log: input has negative values"
    return np.log10(np.clip(x, 1e-8, None))


def from_log(x):
    """10 ** clip(x, max=5)  [pytorch_util.py:161-163]."""
    return np.power(10.0, np.minimum(np.asarray(x), 5.0)).astype(np.asarray(x).dtype)


def wav_to_mel(x, dtype=np.float64):
    """`pre()` of eval_gsr_voicefixer.py:19-25: wav (B,1,L) -> (sp, mel) (B,1,T,1025|128)."""
    sp, _, _ = spectrogram_phase(x, dtype=dtype)
    return sp, mel_project(sp, mel_filterbank(dtype=dtype))
