"""Oracle ResUNet: functional torch-CPU restatement of the reference's conv stacks.

Test infrastructure only (see oracle/__init__.py).  Every function takes the
reference's own ``state_dict`` (keys exactly as ``UNetResComplex_100Mb`` emits
them, e.g. ``encoder_block1.conv_block1.bn1.running_mean``) and evaluates the
eval-mode forward with ``torch.nn.functional`` in NCHW, in whatever dtype the
tensors are (fp32 = the reference's arithmetic, fp64 = ground truth).

Restated reference code:

* ``ConvBlockRes.forward``       models/components/modules.py:263-271 (ctor :223-261)
* ``EncoderBlockRes4B.forward``  models/components/modules.py:178-184
* ``DecoderBlockRes4B.forward``  models/components/modules.py:212-220, ``prune`` :205-210
* mel ``UNetResComplex_100Mb``   models/components/unet.py:60-103
* spectrogram ``UNetResComplex_100Mb`` models/components/unet_v2.py:86-148
* ``Generator.forward``          models/gsr_voicefixer.py:86-91 (log-domain residual)
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-5          # nn.BatchNorm2d default, modules.py:232-233
LRELU_SLOPE = 0.01     # F.leaky_relu_(..., negative_slope=0.01), modules.py:265-266
DOWNSAMPLE = 64        # 2 ** 6 encoder blocks, unet.py:20

ENCODERS = ["encoder_block%d" % i for i in range(1, 7)]
DECODERS = ["decoder_block%d" % i for i in range(1, 7)]


def _bn(sd, p, x):
    """Eval-mode BatchNorm2d: gamma * (x - mean) / sqrt(var + eps) + beta."""
    g, b = sd[p + ".weight"], sd[p + ".bias"]
    m, v = sd[p + ".running_mean"], sd[p + ".running_var"]
    scale = g / torch.sqrt(v + BN_EPS)
    shift = b - m * scale
    return x * scale[None, :, None, None] + shift[None, :, None, None]


def conv_block_res(sd, p, x):
    """Pre-activation residual block (modules.py:263-271).

    Zero padding is applied by conv2d to the *activated* tensor, i.e. the halo is
    0, not lrelu(bn(0)).
    """
    h = F.conv2d(F.leaky_relu(_bn(sd, p + ".bn1", x), LRELU_SLOPE), sd[p + ".conv1.weight"], padding=1)
    h = F.conv2d(F.leaky_relu(_bn(sd, p + ".bn2", h), LRELU_SLOPE), sd[p + ".conv2.weight"], padding=1)
    if (p + ".shortcut.weight") in sd:
        return F.conv2d(x, sd[p + ".shortcut.weight"], sd[p + ".shortcut.bias"]) + h
    return x + h


def encoder_block(sd, p, x):
    """4 residual blocks then floor avg-pool 2x2; returns (pooled, skip)."""
    for i in range(1, 5):
        x = conv_block_res(sd, "%s.conv_block%d" % (p, i), x)
    return F.avg_pool2d(x, kernel_size=2), x


def decoder_block(sd, p, x, skip, both):
    """BN(in_ch) -> ReLU -> ConvTranspose2d(k3, s2, p0) -> prune -> cat -> 4 residual blocks."""
    x = F.conv_transpose2d(F.relu(_bn(sd, p + ".bn1", x)), sd[p + ".conv1.weight"], stride=2)
    x = x[:, :, :-1, :-1] if both else x[:, :, :-1, :]
    x = torch.cat((x, skip), dim=1)
    for i in range(2, 6):
        x = conv_block_res(sd, "%s.conv_block%d" % (p, i), x)
    return x


def _trunk(sd, x, both):
    skips = []
    for p in ENCODERS:
        x, s = encoder_block(sd, p, x)
        skips.append(s)
    x = conv_block_res(sd, "conv_block7", x)
    for p, s in zip(DECODERS, reversed(skips)):
        x = decoder_block(sd, p, x, s, both)
    x = conv_block_res(sd, "after_conv_block1", x)
    return F.conv2d(x, sd["after_conv2.weight"], sd["after_conv2.bias"])


def _pad_time(x):
    T = x.shape[2]
    tpad = int(math.ceil(T / DOWNSAMPLE)) * DOWNSAMPLE
    return F.pad(x, (0, 0, 0, tpad - T)), T


def unet_mel(sd, logmel):
    """Mel-domain ResUNet (unet.py:60-103): (B,1,T,128) -> (B,1,T,128), last bin exactly 0."""
    x, T = _pad_time(logmel)
    x = _trunk(sd, x[..., :-1], both=False)
    return F.pad(x, (0, 1))[:, :, :T, :]


def unet_spec_mag(sd, sp):
    """Trunk of the spectrogram-domain ResUNet (unet_v2.py:100-130): (B,1,T,1025) -> (B,1,T,1025)."""
    x, T = _pad_time(sp)
    x = _trunk(sd, x[..., :-1], both=True)
    return F.pad(x, (0, 1))[:, :, :T, :]


def generator_mel(sd, mel_linear):
    """``Generator.forward`` of models/gsr_voicefixer.py:86-91.

    `sd` holds the analysis module's keys un-prefixed.  mel_linear (B,1,T,128) >= 0.
    Returns the log10-mel estimate.
    """
    assert not bool((mel_linear < 0).any())
    logmel = torch.log10(torch.clamp(mel_linear, min=1e-8))
    return unet_mel(sd, logmel) + logmel
