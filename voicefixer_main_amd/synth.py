"""Deterministic synthetic weights and VCTK-shaped clips (CPU, numpy/torch).

There is no network for checkpoints or datasets, so parity tests and the bench run
on seeded synthetic data of the reference's shapes:

* ResUNet ``state_dict`` with exactly the reference's 660 keys
  (models/components/unet.py:22-53, modules.py:223-261,186-203): Xavier-uniform
  convs as ``init_layer`` (modules.py:276-282) and *non-trivial* BatchNorm
  statistics (the reference's ``init_bn`` gamma=1/beta=0 would hide BN-folding bugs).
* TFGAN vocoder ``state_dict`` for the layer table of ``VocoderSpec``.
* speech-like 44.1 kHz clips: harmonic source (f0 random walk, 30 harmonics, 1/k
  roll-off) x syllabic envelope + noise bursts, degraded like the reference's
  simulator (additive noise SNR~U[-5,40] dB, optional cheby1 low-pass
  tools/dsp/lowpass.py:96-133, optional hard clipping).
"""
from collections import OrderedDict

import numpy as np
import torch

SAMPLE_RATE = 44100

ENC_CHANNELS = [32, 64, 128, 256, 384, 384]                  # unet.py:22-33
DEC_CHANNELS = [(384, 384), (384, 384), (384, 256), (256, 128), (128, 64), (64, 32)]  # unet.py:36-47


# ----------------------------------------------------------------------------
# ResUNet weights
# ----------------------------------------------------------------------------
def resunet_layout(channels_in=1):
    """[(key, shape)] in the reference's state_dict order."""
    out = []

    def bn(p, n):
        out.extend([(p + ".weight", (n,)), (p + ".bias", (n,)), (p + ".running_mean", (n,)),
                    (p + ".running_var", (n,)), (p + ".num_batches_tracked", ())])

    def block(p, cin, cout):
        bn(p + ".bn1", cin)
        bn(p + ".bn2", cout)
        out.append((p + ".conv1.weight", (cout, cin, 3, 3)))
        out.append((p + ".conv2.weight", (cout, cout, 3, 3)))
        if cin != cout:
            out.append((p + ".shortcut.weight", (cout, cin, 1, 1)))
            out.append((p + ".shortcut.bias", (cout,)))

    cin = channels_in
    for i, c in enumerate(ENC_CHANNELS):
        p = "encoder_block%d" % (i + 1)
        block(p + ".conv_block1", cin, c)
        for j in (2, 3, 4):
            block(p + ".conv_block%d" % j, c, c)
        cin = c
    block("conv_block7", 384, 384)
    for i, (ci, co) in enumerate(DEC_CHANNELS):
        p = "decoder_block%d" % (i + 1)
        out.append((p + ".conv1.weight", (ci, co, 3, 3)))
        bn(p + ".bn1", ci)
        block(p + ".conv_block2", 2 * co, co)
        for j in (3, 4, 5):
            block(p + ".conv_block%d" % j, co, co)
    block("after_conv_block1", 32, 32)
    out.append(("after_conv2.weight", (1, 32, 1, 1)))
    out.append(("after_conv2.bias", (1,)))
    return out


def _xavier(shape, gen, gain=1.0):
    rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
    bound = gain * float(np.sqrt(6.0 / ((shape[0] + shape[1]) * rf)))
    return (torch.rand(shape, generator=gen, dtype=torch.float32) * 2.0 - 1.0) * bound


def make_resunet_state_dict(seed=0, channels_in=1, res_gain=1.0):
    """`res_gain` < 1 damps the residual branch of every ConvBlockRes (its conv2) -- the WELL-CONDITIONED network of the
    spectrogram-path parity tests: with every branch at full Xavier gain the 50-block residual stream grows by orders of
    magnitude and the last 1x1 convolution cancels it back down, so two fp32 evaluations of the same trunk agree to ~58 dB
    only and no kernel can be told from another below that; at res_gain = 0.25 the stream stays O(1) and fp32 agrees with
    float64 to > 80 dB (measured in tests/test_oracle_golden.py), so a bar of 70 dB on the split-bf16 kernels is an accuracy
    statement.  The random draws are the same for every gain (same seed -> the same tensors up to that factor)."""
    gen = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for key, shape in resunet_layout(channels_in):
        if key.endswith("num_batches_tracked"):
            sd[key] = torch.tensor(0, dtype=torch.long)
        elif key.endswith("running_var") or (".bn" in key and key.endswith(".weight")):
            sd[key] = torch.rand(shape, generator=gen) + 0.5          # U[0.5, 1.5]
        elif ".bn" in key or key.endswith(".bias"):
            sd[key] = torch.randn(shape, generator=gen) * 0.1          # beta, running_mean, conv biases
        else:
            sd[key] = _xavier(shape, gen)
            if res_gain != 1.0 and key.endswith(".conv2.weight"):
                sd[key] = sd[key] * res_gain
    return sd


# ----------------------------------------------------------------------------
# vocoder weights
# ----------------------------------------------------------------------------
class VocoderSpec:
    """Layer table of the TFGAN generator (unverified recall of the pip package)."""
    n_mels = 128
    cond_channels = 512
    cond_layers = 5
    channels = 1024
    upsample_scales = (7, 7, 3, 3)
    resstack_depth = (8, 8, 8, 8)
    dilation_base = 3
    min_db = -115.0
    amp_floor = 1e-5
    norm_range = 4.0
    up_slope = 0.2
    res_slope = 0.01

    @classmethod
    def tail_frames(cls, T):
        return T % 2 + 4

    @classmethod
    def out_len(cls, T):
        return (T + cls.tail_frames(T)) * int(np.prod(cls.upsample_scales))


def vocoder_layout(spec=VocoderSpec):
    out = []
    cin = spec.n_mels
    for i in range(spec.cond_layers):
        out.append(("condnet.%d" % (2 * i), (spec.cond_channels, cin, 3)))
        cin = spec.cond_channels
    out.append(("generator.1", (spec.channels, cin, 7)))
    c = spec.channels
    idx = 3
    for s, depth in zip(spec.upsample_scales, spec.resstack_depth):
        out.append(("generator.%d.layer" % idx, (c, c // 2, 2 * s)))      # ConvTranspose1d: (Cin, Cout, k)
        c //= 2
        for i in range(depth):
            out.append(("generator.%d.res_layers.%d.1" % (idx + 1, i), (c, c, 3)))
            out.append(("generator.%d.res_layers.%d.3" % (idx + 1, i), (c, c, 3)))
        idx += 3
    out.append(("generator.%d" % (idx + 1), (1, c, 7)))      # 16 for four stages
    return out


def make_vocoder_state_dict(seed=1, spec=VocoderSpec):
    """Variance-controlled random weights: residual branches are damped so the 32-layer
    residual stream stays O(1) and the final tanh is not saturated (a saturated tanh
    would hide numerical error)."""
    gen = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for name, shape in vocoder_layout(spec):
        transposed = name.endswith(".layer")
        fan_in = (shape[0] * 2) if transposed else shape[1] * shape[2]   # 2 taps reach each output phase
        gain = 1.0
        if ".res_layers." in name and name.endswith(".3"):
            gain = 0.25
        if shape[0] == 1:                     # the last k7 convolution
            gain = 0.35
        bound = gain * float(np.sqrt(3.0 / fan_in))
        sd[name + ".weight"] = (torch.rand(shape, generator=gen) * 2.0 - 1.0) * bound
        nb = shape[1] if transposed else shape[0]
        sd[name + ".bias"] = torch.randn((nb,), generator=gen) * 0.02
    return sd


# ----------------------------------------------------------------------------
# clips
# ----------------------------------------------------------------------------
def speech_like(n_samples, seed, sr=SAMPLE_RATE):
    """Clean mono speech-like source in [-1, 1]."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_samples) / sr
    n_ctrl = max(4, int(n_samples / sr * 8) + 2)
    f0_ctrl = np.clip(170.0 + np.cumsum(rng.normal(0, 12.0, n_ctrl)), 90.0, 250.0)
    f0 = np.interp(t, np.linspace(0, t[-1] if n_samples > 1 else 1.0, n_ctrl), f0_ctrl)
    phase = 2.0 * np.pi * np.cumsum(f0) / sr
    x = np.zeros(n_samples)
    tilt = rng.uniform(0.8, 1.4)
    for k in range(1, 31):
        x += np.sin(k * phase + rng.uniform(0, 2 * np.pi)) / (k ** tilt)
    rate = rng.uniform(3.0, 6.0)
    env = np.clip(np.sin(2 * np.pi * rate * t + rng.uniform(0, 6.28)), 0, None) ** 2
    env *= 0.6 + 0.4 * np.sin(2 * np.pi * 0.7 * t + rng.uniform(0, 6.28))
    x *= env
    burst = rng.normal(0, 1.0, n_samples) * (np.clip(np.sin(2 * np.pi * rate * t + 2.1), 0, None) ** 8) * 0.3
    x += np.diff(burst, prepend=0.0)            # high-passed fricative-like bursts
    x /= np.abs(x).max() + 1e-9
    return x


def degrade(x, seed, mode="noise", sr=SAMPLE_RATE):
    """Reference-simulator-like degradations (config/vctk_base_voicefixer_unet.json:80-100)."""
    rng = np.random.default_rng(seed + 7919)
    x = x * rng.uniform(0.3, 0.9)
    snr_db = rng.uniform(-5.0, 40.0)
    p_sig = np.mean(x ** 2) + 1e-12
    noise = rng.normal(0, 1.0, x.shape[0])
    x = x + noise * np.sqrt(p_sig / (10.0 ** (snr_db / 10.0)))
    if mode == "lowpass":
        # the `vctk_cheby1_1000` test set of the reference (evaluation_proc/config.py:91-97): tools/dsp/lowpass.py's
        # lowpass(data, 1000, fs, order=8, _type="cheby1"), mirrored in simulate.py
        from . import simulate
        x = simulate.lowpass(x, 1000, sr, order=8, _type="cheby1")
    elif mode == "clip":
        from . import simulate
        x = simulate.hard_clip(x, 0.25)
    peak = np.abs(x).max()
    if peak > 0.999:
        x = x / peak * 0.999
    return x.astype(np.float32)


def make_clips(n_clips, seconds, seed=1234, mode="noise", distinct=8, sr=SAMPLE_RATE):
    """(n_clips, 1, L) float32 degraded clips.  At most `distinct` clean sources are
    synthesised (they are the expensive part); every clip gets its own gain / noise."""
    L = int(round(seconds * sr))
    bases = [speech_like(L, seed + i, sr) for i in range(min(n_clips, distinct))]
    out = np.empty((n_clips, 1, L), np.float32)
    for i in range(n_clips):
        out[i, 0] = degrade(bases[i % len(bases)], seed + i, mode, sr)
    return out
