#!/usr/bin/env python3
"""Generate tests/golden/* by running the REFERENCE's own modules (imported from
/root/reference, which only exists in the build container) on seeded inputs.

    python oracle/gen_golden.py            # writes tests/golden/*.npz / *.json

Test infrastructure only.  The reference modules that import here are used unmodified:
models/components/{modules,unet,unet_v2}.py, tools/pytorch/mel_scale.py,
tools/pytorch/pytorch_util.py, tools/pytorch/losses.py (table only).  Two sys.modules
stubs make them importable offline: `git` (unet.py:1-6 only asks for the repo root) and
`torchlibrosa.stft` (not installed; unet_v2's FDomainHelper is replaced at run time by a
`torch.stft/istft` helper with the framing of tools/dsp/base.py:52-212, so the spectrogram
ResUNet golden pins the reference TRUNK + phase recombination, not torchlibrosa itself).
"""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


def install_stubs():
    g = types.ModuleType("git")

    class _Git:
        def rev_parse(self, *a, **k):
            return REF

    class Repo:
        def __init__(self, *a, **k):
            self.git = _Git()

    g.Repo = Repo
    sys.modules["git"] = g
    tl, st = types.ModuleType("torchlibrosa"), types.ModuleType("torchlibrosa.stft")

    class _Dummy(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    st.STFT = st.ISTFT = _Dummy
    st.magphase = lambda *a, **k: None
    tl.stft = st
    sys.modules["torchlibrosa"], sys.modules["torchlibrosa.stft"] = tl, st
    pq = types.ModuleType("tools.pytorch.modules.pqmf")     # needs scipy.io .mat files that are not in the repo
    pq.PQMF = _Dummy
    sys.path.insert(0, REF)
    sys.modules.setdefault("tools.pytorch.modules.pqmf", pq)


def install_io_stubs():
    """Empty stand-ins for the I/O and effect packages the reference's DSP / augmentation modules import at the top and the
    functions pinned here never call (librosa, soundfile, progressbar, augment, ...)."""
    import importlib

    class _Any(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return lambda *a, **k: None

    for name in ("librosa", "soundfile", "progressbar", "augment", "sox", "pyroomacoustics", "matplotlib", "matplotlib.pyplot"):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                m = _Any(name)
                m.__all__ = []
                sys.modules[name] = m


class TorchStftHelper(torch.nn.Module):
    """Stand-in for FDomainHelper built on torch.stft/istft (same framing: center, reflect,
    periodic hann, envelope normalisation)."""

    def __init__(self):
        super().__init__()
        self.win = torch.hann_window(2048, periodic=True)

    def wav_to_spectrogram_phase(self, wav, eps=1e-8):
        B, C, L = wav.shape
        s = torch.stft(wav.reshape(B * C, L), 2048, 441, 2048, self.win, center=True, pad_mode="reflect",
                       return_complex=True).transpose(1, 2)
        re, im = s.real, s.imag
        mag = torch.clamp(re ** 2 + im ** 2, eps, np.inf) ** 0.5
        shp = (B, C) + tuple(mag.shape[1:])
        return mag.reshape(shp), (re / mag).reshape(shp), (im / mag).reshape(shp)

    def istft(self, real, imag, length):
        spec = torch.complex(real[:, 0], imag[:, 0]).transpose(1, 2)
        spec = spec.clone()
        spec[:, 0].imag.zero_()
        spec[:, -1].imag.zero_()
        return torch.istft(spec, 2048, 441, 2048, self.win, center=True, length=length)


BANDPASS_CASES = {"d_butter": (500, 4000, 5, "butter"), "d_cheby1": (300.7, 3400.2, 6, "cheby1"), "d_ellip": (1000, 8000, 4, "ellip"),
                  "d_bessel": (200, 2000, 3, "bessel"), "d_substr": (400, 5000, 5, "utt"), "d_clamped": (400, 5000, 13, "cheby1")}


def simulate_golden():
    """Section 7 alone: `python oracle/gen_golden.py --only simulate` rewrites tests/golden/simulate.npz and nothing else."""
    install_stubs()
    # 7. degradation simulator (round 4): the reference's OWN tools/dsp/lowpass.py and
    # dataloaders/augmentation/base.py on seeded inputs -- every IIR type, the resampling low-pass, the substring dispatch, the
    # band-pass, the three noise mixers with the random draws pinned (equal bounds: tools/pytorch/random_.py returns the bound).
    # `stft_hard` needs torchlibrosa (FDomainHelper) and is covered by the GPU test against the oracle's STFT instead.
    install_io_stubs()
    import tools.dsp.lowpass as ref_lp
    import dataloaders.augmentation.base as ref_aug
    rng = np.random.default_rng(31)
    n = 4096
    x = rng.normal(0.0, 0.1, n)
    sim = {"x": x}
    for name, (hc, order, typ) in {"butter": (4000, 5, "butter"), "cheby1": (1000, 8, "cheby1"), "ellip": (6000, 6, "ellip"),
                                   "bessel": (2000, 4, "bessel"), "substr_b": (3000, 5, "b"), "order_clamped": (3000, 14, "butter"),
                                   "stft": (8000, 5, "stft")}.items():
        sim["lowpass_" + name] = np.asarray(ref_lp.lowpass(x.copy(), highcut=hc, fs=44100, order=order, _type=typ), np.float64)
    sim["bandpass_butter"] = np.asarray(ref_lp.bandpass_filter(x.copy(), 500, 4000, 44100, 6, "butter"), np.float64)
    sim["bandpass_cheby2"] = np.asarray(ref_lp.bandpass_filter(x.copy(), 300, 3000, 44100, 4, "cheby2"), np.float64)
    # the dispatch wrapper (lowpass.py:189-215; round 5): every type it accepts, a substring, a clamped order, float cut-offs
    for name, (lc, hc, order, typ) in BANDPASS_CASES.items():
        sim["bandpass_" + name] = np.asarray(ref_lp.bandpass(x.copy(), lc, hc, 44100, order=order, _type=typ), np.float64)
    front, noise = rng.normal(0.0, 0.2, n), rng.normal(0.0, 0.05, n)
    hq, aug = front * 0.9 + rng.normal(0.0, 0.01, n), np.tanh(front * 3.0)
    sim.update(front=front, noise=noise, hq=hq, aug=aug)
    t = lambda a: torch.from_numpy(a.copy())
    o = ref_aug.add_noise_and_scale(t(front), t(noise), snr_l=10, snr_h=10, scale_lower=0.8, scale_upper=0.8)
    sim["mix_front"], sim["mix_noise"], sim["mix_snr_scale"] = o[0].numpy(), o[1].numpy(), np.array([float(o[2]), float(o[3])])
    o = ref_aug.add_noise_and_scale_with_HQ(t(hq), t(front), t(noise), snr_l=5, snr_h=5, scale_lower=0.7, scale_upper=0.7)
    sim["hq_hq"], sim["hq_front"], sim["hq_noise"], sim["hq_snr_scale"] = o[0].numpy(), o[1].numpy(), o[2].numpy(), np.array([float(o[3]), float(o[4])])
    o = ref_aug.add_noise_and_scale_with_HQ_with_Aug(t(hq), t(front), t(aug), t(noise), snr_l=0, snr_h=0, scale_lower=0.9, scale_upper=0.9)
    sim["aug_hq"], sim["aug_front"], sim["aug_aug"], sim["aug_noise"], sim["aug_snr_scale"] = (
        o[0].numpy(), o[1].numpy(), o[2].numpy(), o[3].numpy(), np.array([float(o[4]), float(o[5])]))
    np.savez_compressed(os.path.join(OUT, "simulate.npz"), **{k: np.asarray(v, np.float64) for k, v in sim.items()})


def main():
    install_stubs()
    os.makedirs(OUT, exist_ok=True)
    from voicefixer_main_amd import synth

    from models.components.unet import UNetResComplex_100Mb as MelUNet
    from tools.pytorch.mel_scale import MelScale
    from tools.pytorch.pytorch_util import from_log, to_log
    from tools.pytorch.losses import mel_weight_44k_128

    # 1. state_dict layout of the reference module
    ref = MelUNet(channels=1).eval()
    layout = [[k, list(v.shape)] for k, v in ref.state_dict().items()]
    json.dump(layout, open(os.path.join(OUT, "resunet_layout.json"), "w"))

    # 2. mel filterbank buffer (sparse) and the vocoder band-weight table
    fb = MelScale(n_mels=128, sample_rate=44100, n_stft=1025).fb.numpy()
    nz = np.nonzero(fb)
    np.savez_compressed(os.path.join(OUT, "mel_fb.npz"), rows=nz[0].astype(np.int16), cols=nz[1].astype(np.int16),
                        vals=fb[nz], band_weight=mel_weight_44k_128.numpy().ravel())

    # 3. mel ResUNet + Generator.forward composition, synthetic weights seed 0
    sd = synth.make_resunet_state_dict(0)
    ref.load_state_dict(sd)
    g = torch.Generator().manual_seed(7)
    mel = 10.0 ** (torch.randn((2, 1, 101, 128), generator=g) * 1.2 - 2.5)
    with torch.no_grad():
        lg = to_log(mel)
        unet_out = ref(lg)["mel"]
        out = unet_out + lg                                    # models/gsr_voicefixer.py:86-91
        lin = from_log(out)
    np.savez_compressed(os.path.join(OUT, "unet_mel.npz"), mel_in=mel.numpy(), unet_out=unet_out.numpy(),
                        logmel_out=out.numpy(), from_log_checksum=np.float64(lin.double().sum().item()))

    # 4. one ConvBlockRes / encoder / decoder block in isolation (small tensors)
    from models.components.modules import ConvBlockRes, DecoderBlockRes4B, EncoderBlockRes4B
    torch.manual_seed(11)
    blk = ConvBlockRes(32, 64, (3, 3), "relu", 0.01).eval()
    enc = EncoderBlockRes4B(32, 32, (2, 2), "relu", 0.01).eval()
    dec = DecoderBlockRes4B(64, 32, (2, 2), "relu", 0.01).eval()
    gen = torch.Generator().manual_seed(12)
    for m in (blk, enc, dec):
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.data = torch.rand(mod.num_features, generator=gen) + 0.5
                mod.bias.data = torch.randn(mod.num_features, generator=gen) * 0.1
                mod.running_mean.data = torch.randn(mod.num_features, generator=gen) * 0.1
                mod.running_var.data = torch.rand(mod.num_features, generator=gen) + 0.5
    x = torch.randn((1, 32, 6, 7), generator=gen)
    skip = torch.randn((1, 32, 6, 7), generator=gen)
    xd = torch.randn((1, 64, 3, 3), generator=gen)
    with torch.no_grad():
        yb = blk(x)
        yp, ye = enc(x)
        yd = dec(xd, skip)
        yd_both = dec(xd, skip[..., :6], both=True)
    blocks = {"x": x.numpy(), "skip": skip.numpy(), "xd": xd.numpy(), "block": yb.numpy(), "enc_pool": yp.numpy(),
              "enc": ye.numpy(), "dec": yd.numpy(), "dec_both": yd_both.numpy()}
    for name, m in (("blk", blk), ("enc", enc), ("dec", dec)):
        for k, v in m.state_dict().items():
            if not k.endswith("num_batches_tracked"):
                blocks["%s/%s" % (name, k)] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "blocks.npz"), **blocks)

    # 5. spectrogram ResUNet (unet_v2): reference trunk + recombination, torch.stft helper
    from models.components import unet_v2
    spec = unet_v2.UNetResComplex_100Mb(channels=1).eval()
    spec.f_helper = TorchStftHelper()
    sd2 = synth.make_resunet_state_dict(2)
    spec.load_state_dict(sd2, strict=False)
    # L = 14312 = 32 * 441 + 200: NOT a multiple of the hop, so the fixture pins the last L mod 441 samples, which
    # torch.istft(length=L) (like torchlibrosa and tools/dsp/base.py:193-200) reconstructs from the last frames' tails
    wav = torch.from_numpy(synth.make_clips(1, 14312 / 44100.0, seed=99))   # (1,1,14312) -> T = 33
    sp, _, _ = spec.f_helper.wav_to_spectrogram_phase(wav)
    mags = {}
    spec.after_conv2.register_forward_hook(lambda m, i, o: mags.__setitem__("mag", o.detach()))
    with torch.no_grad():
        out_wav = spec(sp, wav)["wav"]
    np.savez_compressed(os.path.join(OUT, "unet_spec.npz"), wav_in=wav.numpy(), wav_out=out_wav.numpy(),
                        mag_sub=mags["mag"][0, 0, ::4, ::16].numpy())

    # 6. the same reference module on a BATCH of two longer clips (round 4): L = 129 * 441 + 200 -> T = 130 frames, Tpad = 192 --
    # the zero padding of the time axis crosses a 64-frame boundary that is not the first one, every level of the trunk has
    # more than one tile row, and clip 1 sits behind clip 0 in every tensor.  FULL-BAND clips (additive noise, no low-pass):
    # on a low-passed clip the phase of the numerically empty bins above the cut-off is rounding noise / 1e-4 (the clamp of
    # fDomainHelper.py:60-65), different noise in every implementation, which bounds ANY two fp32 evaluations at ~60 dB
    # (scripts/ssr_conditioning.py); here two evaluations agree to > 100 dB, so the bars of the GPU tests are accuracy
    # statements.  The input is PCM16-quantised so that the fixture stores it exactly in 2 bytes per sample.
    L2 = 129 * 441 + 200
    pcm = np.round(synth.make_clips(2, L2 / 44100.0, seed=4242, mode="noise")[:, 0] * 32767.0).astype(np.int16)
    wav2 = torch.from_numpy(pcm.astype(np.float32) / 32768.0)[:, None]
    sp2, _, _ = spec.f_helper.wav_to_spectrogram_phase(wav2)
    with torch.no_grad():
        out2 = spec(sp2, wav2)["wav"]
    np.savez_compressed(os.path.join(OUT, "unet_spec_b2.npz"), pcm_in=pcm, wav_out=out2.numpy()[:, 0],
                        mag_sub=mags["mag"][:, 0, ::8, ::32].numpy())
    simulate_golden()
    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print("  %-24s %8d bytes" % (f, os.path.getsize(os.path.join(OUT, f))))


if __name__ == "__main__":
    if sys.argv[1:] == ["--only", "simulate"]:
        simulate_golden()
    else:
        main()
