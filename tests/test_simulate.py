"""CPU: the degradation simulator (voicefixer_main_amd/simulate.py = tools/dsp/lowpass.py + dataloaders/augmentation/base.py of the
reference) against SciPy called directly, plus the properties a degradation must have.  The reference
file itself cannot be imported here (librosa, torchlibrosa); its `stft_hard` branch is the `-m gpu` test at the bottom."""
import os
import sys

import numpy as np
import pytest
from scipy import signal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from voicefixer_main_amd import simulate, synth  # noqa: E402

FS = 44100


def _noise(n=44100, seed=0):
    return np.random.default_rng(seed).normal(0, 0.1, n)


def _band_power_db(x, lo, hi):
    f, p = signal.welch(x, FS, nperseg=4096)
    return 10 * np.log10(p[(f >= lo) & (f < hi)].mean() + 1e-30)


@pytest.mark.parametrize("ftype,design", [
    ("butter", lambda o, w: signal.butter(o, w, btype="low", output="sos")),
    ("cheby1", lambda o, w: signal.cheby1(o, 0.1, w, btype="low", output="sos")),
    ("ellip", lambda o, w: signal.ellip(o, 0.1, 60, w, btype="low", output="sos")),
    ("bessel", lambda o, w: signal.bessel(o, w, btype="low", output="sos")),
])
def test_iir_lowpass_equals_scipy_called_directly(ftype, design):
    x = _noise()
    for highcut, order in ((1000, 8), (4000.7, 5), (8000, 2)):
        want = signal.sosfiltfilt(design(order, int(highcut) / (0.5 * FS)), x)        # lowpass.py:96-133: highcut = int(highcut)
        got = simulate.lowpass(x, highcut, FS, order=order, _type=ftype)
        assert got.shape == x.shape and np.array_equal(got, want), (ftype, highcut, order)
    # lowpass.py:148-165: the order is limited to [2, 10] and truncated
    assert np.array_equal(simulate.lowpass(x, 2000, FS, order=40, _type=ftype), simulate.lowpass(x, 2000, FS, order=10, _type=ftype))
    assert np.array_equal(simulate.lowpass(x, 2000, FS, order=1, _type=ftype), simulate.lowpass(x, 2000, FS, order=2, _type=ftype))
    assert np.array_equal(simulate.lowpass(x, 2000, FS, order=5.9, _type=ftype), simulate.lowpass(x, 2000, FS, order=5, _type=ftype))
    # a degradation: the pass band stays, the stop band goes
    y = simulate.lowpass(x, 2000, FS, order=8, _type=ftype)
    assert abs(_band_power_db(y, 100, 500) - _band_power_db(x, 100, 500)) < (3.0 if ftype == "bessel" else 1.0)   # (Bessel droops early, twice)
    assert _band_power_db(y, 8000, 20000) < _band_power_db(x, 8000, 20000) - (25 if ftype == "bessel" else 55)


def test_lowpass_dispatch_follows_the_reference():
    x = _noise(8000)
    with pytest.raises(ValueError):
        simulate.lowpass(x[:, None], 1000, FS)                                     # (samples, 1) is refused (lowpass.py:164-165)
    with pytest.raises(ValueError):
        simulate.lowpass(x, 1000, FS, _type="chebyshev")
    # `_type in "butter"` is a substring test in the reference: kept
    b = simulate.lowpass(x, 1000, FS, order=5, _type="butter")
    assert np.array_equal(simulate.lowpass(x, 1000, FS, order=5, _type="butt"), b)
    assert np.array_equal(simulate.lowpass(x, 1000, FS, order=5, _type=""), b)
    assert np.array_equal(simulate.lowpass(x, 1000, FS, order=5, _type="e"), b)    # "e" is found in "butter" before "ellip" is asked
    assert np.array_equal(simulate.lowpass(x, 1000, FS), b)                         # defaults: order 5, butter
    with pytest.raises(Exception):
        simulate.lowpass_filter(x, 1000, FS, 4, "fir")
    assert simulate.limit(11, 10, 2) == 10 and simulate.limit(1, 10, 2) == 2 and simulate.limit(4.7, 10, 2) == 4
    assert len(simulate.align_length(np.zeros(10), np.ones(7))) == 10 and simulate.align_length(np.zeros(10), np.ones(7))[7:].sum() == 0
    assert len(simulate.align_length(np.zeros(5), np.ones(7))) == 5


def test_resampling_lowpass_equals_scipy_called_directly():
    """`_type="stft"` (lowpass.py:135-146) = resample_poly down to int(ratio * 44100) and back up, length aligned."""
    x = _noise(30001, 3)
    for highcut in (1000, 4000, 11025):
        ratio = highcut / int(FS / 2)
        fs_down = int(ratio * 44100)
        want = signal.resample_poly(signal.resample_poly(x, fs_down, 44100), 44100, fs_down)
        want = np.pad(want, (0, max(0, len(x) - len(want))))[:len(x)]
        got = simulate.lowpass(x, highcut, FS, _type="stft")
        assert got.shape == x.shape and np.array_equal(got, want), highcut
        assert _band_power_db(got, 1.5 * highcut, 22050) < _band_power_db(x, 1.5 * highcut, 22050) - 40 or highcut > 9000


def test_bandpass_filter_equals_scipy():
    x = _noise(20000, 5)
    for ftype, sos in (("butter", signal.butter(4, [300 / 22050, 3400 / 22050], btype="band", output="sos")),
                       ("cheby2", signal.cheby2(4, 60, [300 / 22050, 3400 / 22050], btype="band", output="sos"))):
        assert np.array_equal(simulate.bandpass_filter(x, 300, 3400, FS, 4, ftype), signal.sosfiltfilt(sos, x))


def test_synthetic_test_set_uses_the_simulator():
    """synth.degrade(mode="lowpass") IS lowpass(x, 1000, 44100, order=8, _type="cheby1") -- the reference's `vctk_cheby1_1000`
    set -- bit for bit what rounds 1-3 generated with the hard-wired filter (the committed goldens depend on it)."""
    base = synth.speech_like(20000, 1234)
    rng = np.random.default_rng(1234 + 7919)
    x = base * rng.uniform(0.3, 0.9)
    snr_db = rng.uniform(-5.0, 40.0)
    x = x + rng.normal(0, 1.0, x.shape[0]) * np.sqrt((np.mean(x ** 2) + 1e-12) / (10.0 ** (snr_db / 10.0)))
    want = signal.sosfiltfilt(signal.cheby1(8, 0.1, 1000.0 / (FS / 2), btype="low", output="sos"), x)
    peak = np.abs(want).max()
    want = (want / peak * 0.999 if peak > 0.999 else want).astype(np.float32)
    assert np.array_equal(synth.degrade(base, 1234, "lowpass"), want)
    c = synth.degrade(base, 1234, "clip")
    assert np.abs(c).max() <= 0.25 + 1e-7 and (np.abs(c) > 0.2499).sum() > 10


def test_snr_mixing_follows_the_reference():
    """dataloaders/augmentation/base.py:33-118: peak-normalised signals, the noise lowered by snr dB (amplitude ratio), the
    mixture's peak to 1, one common random scale."""
    rng = np.random.default_rng(0)
    front, noise = _noise(5000, 1) * 3.0, _noise(5000, 2) * 0.01
    f, n, snr, scale = simulate.add_noise_and_scale(front, noise, snr_l=10, snr_h=10.000001, scale_lower=0.5, scale_upper=0.5, rng=rng)
    assert snr == 10.000001 and scale == 0.5                                        # (almost) empty intervals return their upper bound
    assert abs(np.abs(f + n).max() - 0.5) < 1e-12                                    # mixture peak 1, then the scale
    assert abs(20 * np.log10(np.abs(f).max() / np.abs(n).max()) - snr) < 1e-6       # peak ratio = the SNR
    assert np.allclose(f / np.abs(f).max(), front / np.abs(front).max())             # only scaled
    f, n, snr, scale = simulate.add_noise_and_scale(front, noise, rng=np.random.default_rng(1))
    assert -5 <= snr < 35 and 0.6 <= scale < 1.0 and abs(np.abs(f + n).max() - scale) < 1e-12
    f2, n2, snr2, _ = simulate.add_noise_and_scale(front, noise, snr_l=None, snr_h=None, rng=rng)
    assert snr2 is None and abs(np.abs(f2).max() / np.abs(n2).max() - 1.0) < 1e-9    # no SNR: both at unit peak before the mix
    # with_HQ: HQ and front share one factor; the noise level follows the speech unless it is nearly silent
    HQ = front * 1.0
    aug = np.clip(front, -1.0, 1.0)
    h, f, a, n, snr, scale = simulate.add_noise_and_scale_with_HQ_with_Aug(HQ, front, aug, noise, snr_l=0, snr_h=0, scale_lower=1, scale_upper=1, rng=rng)
    assert snr == 0 and scale == 1 and np.allclose(h, f) and abs(max(np.abs(a + n).max(), np.abs(h).max()) - 1.0) < 1e-9
    assert abs(np.mean(np.abs(n)) / np.mean(np.abs(a)) - 1.0) < 1e-6                 # level-matched at 0 dB
    h, f, n, snr, scale = simulate.add_noise_and_scale_with_HQ(HQ, front, noise, snr_l=20, snr_h=20, scale_lower=1, scale_upper=1, rng=rng)
    assert abs(np.mean(np.abs(f)) / np.mean(np.abs(n)) - 10.0) < 1e-6                # 20 dB below the speech level
    quiet = front * 1e-9 / np.abs(front).max()
    assert simulate._match_noise_level(noise, quiet) is noise                        # nearly silent speech: the noise is left alone


@pytest.mark.gpu
def test_stft_hard_lowpass_on_the_device_vs_oracle():
    """`_type="stft_hard"` (lowpass.py:21-33) runs on the GPU (front-end kernel in its phase-emitting form, mask, ISTFT kernel):
    against the float64 oracle's STFT -> mask -> ISTFT, at an unaligned length, and as a degradation (nothing above the cut)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import dsp
    from voicefixer_main_amd.engine import Engine
    eng = Engine("cuda:0")
    x = synth.make_clips(1, (60 * 441 + 123) / 44100.0, seed=31)[0, 0]
    for highcut in (1000, 4000, 12000):
        ratio = highcut / int(FS / 2)
        got = simulate.lowpass(x, highcut, FS, _type="stft_hard", engine=eng)
        assert got.dtype == np.float32 and got.shape == x.shape
        mag, cos, sin = dsp.spectrogram_phase(x[None, None].astype(np.float64), dtype=np.float64)
        cut = int(1025 * ratio)
        mag[..., cut:] = 0.0
        ref = dsp.istft((mag * cos)[0], (mag * sin)[0], x.shape[0], dtype=np.float64)[0]
        assert np.abs(got - ref).max() < 2e-5 * max(1.0, np.abs(ref).max()), (highcut, np.abs(got - ref).max())
        assert _band_power_db(got, 1.2 * highcut + 200, 22050) < _band_power_db(x, 1.2 * highcut + 200, 22050) - 30
    simulate.set_engine(eng)
    assert np.array_equal(simulate.lowpass(x, 4000, FS, _type="stft_hard"), simulate.lowpass(x, 4000, FS, _type="stft_hard", engine=eng))
    assert eng.take_flags() == 0
