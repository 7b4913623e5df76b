"""CPU emulation of the vocoder's 16-bit mode with an fp16 RESIDUAL TRUNK (round-4 gate study).

float64 arithmetic everywhere; what is emulated is WHERE values are rounded to fp16:
  operands : every conv input (after its activation) and every weight            (= precision 2 today)
  trunk    : additionally the ResStack trunk after every layer and every upsampler output
Reports SI-SDR of each variant against the un-rounded float64 forward, damped and un-damped weights.
Usage: python scripts/fp16_trunk_emulation.py [seconds]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from oracle import vocoder as ov, dsp
from voicefixer_main_amd import synth


def r16(x):
    return x.to(torch.float16).to(torch.float64)


def gen(sd, cond, cfg, operands, trunk, pair_keep=False):
    q = r16 if operands else (lambda t: t)
    tq = r16 if trunk else (lambda t: t)
    W = {k: (q(v.double()) if k.endswith("weight") else v.double()) for k, v in sd.items()}
    x = cond
    for i in range(cfg.cond_layers):
        x = F.elu(F.conv1d(q(x), W["condnet.%d.weight" % (2 * i)], W["condnet.%d.bias" % (2 * i)], padding=1))
    x = F.conv1d(F.pad(q(x), (3, 3), mode="reflect"), W["generator.1.weight"], W["generator.1.bias"])
    idx = 3
    for s, depth in zip(cfg.upsample_scales, cfg.resstack_depth):
        x = q(F.leaky_relu(x, cfg.up_slope))
        x = F.conv_transpose1d(x, W["generator.%d.layer.weight" % idx], W["generator.%d.layer.bias" % idx],
                               stride=s, padding=s // 2 + s % 2, output_padding=s % 2)
        x = tq(x)
        p = "generator.%d" % (idx + 1)
        for i in range(depth):
            d = cfg.dilation_base ** i
            h = F.conv1d(q(F.leaky_relu(x, cfg.res_slope)), W["%s.res_layers.%d.1.weight" % (p, i)],
                         W["%s.res_layers.%d.1.bias" % (p, i)], padding=d, dilation=d)
            h = F.conv1d(q(F.leaky_relu(h, cfg.res_slope)), W["%s.res_layers.%d.3.weight" % (p, i)],
                         W["%s.res_layers.%d.3.bias" % (p, i)], padding=1)
            x = x + h
            if not (pair_keep and i in (0, 2)):   # first layer of a fused pair stays fp32 on chip
                x = tq(x)
        idx += 3
    x = F.leaky_relu(x, cfg.up_slope)
    x = F.conv1d(F.pad(x, (3, 3), mode="reflect"), W["generator.16.weight"], W["generator.16.bias"])
    return torch.tanh(x)


def sisdr(ref, est):
    e = est - ref
    return float(10 * np.log10((ref ** 2).sum() / ((e ** 2).sum() + 1e-300)))


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 0.4
    cfg = ov.VocoderConfig()
    wav = synth.make_clips(2, secs)
    mel = dsp.mel_from_wav(torch.from_numpy(wav).double()) if hasattr(dsp, "mel_from_wav") else None
    if mel is None:
        sp = dsp.stft_mag(wav[:, 0].astype(np.float64)) if hasattr(dsp, "stft_mag") else None
    for damp in (True, False):
        sd = synth.make_vocoder_state_dict(1)
        if not damp:
            for k in sd:
                if ".res_layers." in k and k.endswith(".3.weight"):
                    sd[k] = sd[k] * 4.0
        torch.manual_seed(0)
        T = int(secs * 100) + 1
        melx = torch.rand(2, 1, T, 128, dtype=torch.float64) ** 4 * 3.0     # linear mel with a wide dynamic range
        cond = ov.normalise_mel(melx, cfg)
        ref = gen(sd, cond, cfg, False, False).numpy()
        for name, o, t, pk in (("operands fp16 (today)", True, False, False), ("operands + trunk fp16", True, True, False),
                               ("operands + trunk fp16, pairs keep y1", True, True, True), ("trunk fp16 only", False, True, False)):
            est = gen(sd, cond, cfg, o, t, pk).numpy()
            print("damped=%s  %-40s SI-SDR %.1f dB  max err %.2e (peak %.2f)" % (damp, name, sisdr(ref, est), np.abs(est - ref).max(), np.abs(ref).max()), flush=True)


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
