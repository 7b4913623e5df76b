"""Oracle TFGAN vocoder: torch-CPU restatement of the third-party `voicefixer` Vocoder.

Test infrastructure only (see oracle/__init__.py).

PARITY UNPINNED.  The vocoder the reference calls (`from voicefixer import Vocoder`,
models/gsr_voicefixer.py:4,113; call sites eval_gsr_voicefixer.py:66,
models/gsr_voicefixer.py:272) lives in the un-pinned PyPI package `voicefixer`
(requirements.txt:6).  Neither its source nor its pretrained 44.1 kHz checkpoint is
in /root/reference or installable offline.  This file restates the published
structure as recorded in SURVEY.md §8c ("recalled, unverified"), driven entirely by
a layer table (`VocoderConfig`) so that a different table can be dropped in once the
real package can be inspected.  The only in-repo anchors are the band-weight table
`mel_weight_44k_128` (tools/pytorch/losses.py:17-50) used as the input normaliser
and the (B,1,T,n_mel) input contract (tools/pytorch/vocoder_loss.py:48-93).

Structure restated:
  Vocoder.__call__ : mel / band_weight -> 20*log10(max(|.|,1e-5)) - 20 -> affine map of
                     [min_db, 0] onto [-4, 4] with clipping -> (B,128,T) -> append
                     (T % 2 + 4) frames of -4.0
  Generator        : condnet 5 x [Conv1d k3 p1 + ELU] (128->512->512...) ;
                     ReflectionPad1d(3) + Conv1d k7 (512->1024) ;
                     4 x { LeakyReLU(0.2) ; ConvTranspose1d(k=2s, stride s,
                           padding s//2 + s%2, output_padding s%2) halving channels,
                           s = 7,7,3,3 ; ResStack(depth 8): x += Conv1d_k3_d1(
                           LeakyReLU(0.01)(Conv1d_k3_dil(3**i)(LeakyReLU(0.01)(x)))) } ;
                     LeakyReLU(0.2) ; ReflectionPad1d(3) + Conv1d k7 (64->1) ; tanh
  Weight-norm is assumed folded into plain weights at load time.

State-dict key convention (mirrors nn.Sequential indexing of the package):
  condnet.{0,2,4,6,8}.{weight,bias}
  generator.1.{weight,bias}                       first k7 conv
  generator.{3,6,9,12}.layer.{weight,bias}        transposed-conv upsamplers
  generator.{4,7,10,13}.res_layers.{i}.{1,3}.{weight,bias}
  generator.{3 n + 4}.{weight,bias}               last k7 conv (n stages; 16 for the table above)
"""
from dataclasses import dataclass, field
from typing import List

import numpy as np
import torch
import torch.nn.functional as F

# tools/pytorch/losses.py:17-50 divides this table by its first element; the values
# are the mel-band widths of the 128-band HTK scale, i.e. get_mel_weig()
# (tools/pytorch/pytorch_util.py:141-155) with base 10.  We rebuild them from that
# formula (checked against the table in tests).
def mel_band_weight(n_mel=128, sample_rate=44100):
    alpha = 2595.0
    m_max = alpha * np.log10(1.0 + (sample_rate // 2) / 700.0)
    m_pts = np.linspace(0.0, m_max, n_mel + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / alpha) - 1.0)
    norm = (f_pts[2:] - f_pts[:-2]) / 2.0
    return (norm / norm[0]).astype(np.float32)


@dataclass
class VocoderConfig:
    n_mels: int = 128
    cond_channels: int = 512
    cond_layers: int = 5
    channels: int = 1024
    upsample_scales: List[int] = field(default_factory=lambda: [7, 7, 3, 3])
    resstack_depth: List[int] = field(default_factory=lambda: [8, 8, 8, 8])
    res_kernel: int = 3
    dilation_base: int = 3
    min_db: float = -115.0
    amp_floor: float = 1e-5
    norm_range: float = 4.0
    up_slope: float = 0.2      # LeakyReLU in front of every upsampler / the last conv
    res_slope: float = 0.01    # nn.LeakyReLU() default inside the ResStack

    def hop(self):
        return int(np.prod(self.upsample_scales))

    def tail_frames(self, T):
        return T % 2 + 4


def normalise_mel(mel, cfg=VocoderConfig()):
    """(B,1,T,128) linear mel -> (B,128,T+tail) conditioning in [-4, 4]."""
    w = torch.as_tensor(mel_band_weight(cfg.n_mels), dtype=mel.dtype)
    s = 20.0 * torch.log10(torch.clamp(torch.abs(mel / w), min=cfg.amp_floor)) - 20.0
    s = torch.clamp((s - cfg.min_db) / (-cfg.min_db) * (2 * cfg.norm_range) - cfg.norm_range,
                    -cfg.norm_range, cfg.norm_range)
    x = s[:, 0].transpose(1, 2)
    tail = torch.full((x.shape[0], x.shape[1], cfg.tail_frames(x.shape[2])), -cfg.norm_range, dtype=x.dtype)
    return torch.cat((x, tail), dim=2)


def _res_stack(sd, p, x, depth, cfg):
    for i in range(depth):
        d = cfg.dilation_base ** i
        h = F.conv1d(F.leaky_relu(x, cfg.res_slope), sd["%s.res_layers.%d.1.weight" % (p, i)],
                     sd["%s.res_layers.%d.1.bias" % (p, i)], padding=d, dilation=d)
        h = F.conv1d(F.leaky_relu(h, cfg.res_slope), sd["%s.res_layers.%d.3.weight" % (p, i)],
                     sd["%s.res_layers.%d.3.bias" % (p, i)], padding=1)
        x = x + h
    return x


def generator(sd, cond, cfg=VocoderConfig()):
    """(B,128,T') conditioning -> (B,1,T'*441) waveform."""
    x = cond
    for i in range(cfg.cond_layers):
        x = F.elu(F.conv1d(x, sd["condnet.%d.weight" % (2 * i)], sd["condnet.%d.bias" % (2 * i)], padding=1))
    x = F.conv1d(F.pad(x, (3, 3), mode="reflect"), sd["generator.1.weight"], sd["generator.1.bias"])
    idx = 3
    for s, depth in zip(cfg.upsample_scales, cfg.resstack_depth):
        x = F.leaky_relu(x, cfg.up_slope)
        x = F.conv_transpose1d(x, sd["generator.%d.layer.weight" % idx], sd["generator.%d.layer.bias" % idx],
                               stride=s, padding=s // 2 + s % 2, output_padding=s % 2)
        x = _res_stack(sd, "generator.%d" % (idx + 1), x, depth, cfg)
        idx += 3
    x = F.leaky_relu(x, cfg.up_slope)
    last = "generator.%d" % (idx + 1)          # nn.Sequential indexing: activation at idx - 1, pad at idx, conv at idx + 1 (16 for 4 stages)
    x = F.conv1d(F.pad(x, (3, 3), mode="reflect"), sd[last + ".weight"], sd[last + ".bias"])
    return torch.tanh(x)


def vocoder(sd, mel_linear, cfg=VocoderConfig()):
    """`model.vocoder(mel)`: (B,1,T,128) linear mel -> (B,1,(T + T%2 + 4) * 441)."""
    assert mel_linear.shape[-1] == cfg.n_mels
    return generator(sd, normalise_mel(mel_linear, cfg), cfg)
