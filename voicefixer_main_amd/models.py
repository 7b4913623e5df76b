"""Host-side mirror of the reference's model objects for the inference hot path.

Same names, argument meaning and error behaviour as the reference surface that
``handler()`` uses, backed by libvfx.so instead of torch.nn:

    model = VoiceFixer(hp, channels=2, type_target="vocals").load_from_checkpoint(ckpt)
    model.eval(); model = model.to(device)
    sp, cos, sin = model.f_helper.wav_to_spectrogram_phase(wav)      # fDomainHelper.py:67-89
    mel = model.mel(sp.permute(0,1,3,2)).permute(0,1,3,2)            # mel_scale.py:52-64
    out = model(mel)['mel']                                          # gsr_voicefixer.py:183-193
    wav = model.vocoder(from_log(out))                               # eval_gsr_voicefixer.py:66

and for the waveform-out ResUNets (models/ssr_unet.py:145-155, models/gsr_unet.py):

    out = model(sp, wav)   ->  {'wav': (B,1,L), 'clean': sp}

Tensors are torch tensors on the ROCm device; the batch dimension may be > 1 (the
reference only ever passes 1).  There is no CPU path: constructing a model on a
non-GPU device raises.
"""
import math
import pickle

import numpy as np
import torch

from .engine import Engine, MODEL_UNET_MEL, MODEL_UNET_SPEC, MODEL_VOCODER

EPS = 1e-8


# ----------------------------------------------------------------------------------------
# tools/pytorch/pytorch_util.py:157-163 (elementwise glue, kept in torch on the device)
# ----------------------------------------------------------------------------------------
def to_log(input):
    assert torch.sum(input < 0) == 0, str(input) + " has negative values counts " + str(torch.sum(input < 0))
    return torch.log10(torch.clip(input, min=1e-8))


def from_log(input):
    return 10 ** torch.clip(input, max=5)


def tensor2numpy(tensor):
    return tensor.detach().cpu().numpy()


def _hp_get(hp, *keys, default=None):
    cur = hp
    for k in keys:
        try:
            cur = cur[k]
        except Exception:
            cur = getattr(cur, k, None)
        if cur is None:
            return default
    return cur


class MelScale:
    """tools/pytorch/mel_scale.py:10-64 -- HTK triangular filterbank applied to (..., freq, time)."""

    def __init__(self, engine, n_mels=128, sample_rate=44100, f_min=0.0, f_max=None, n_stft=1025, norm=None,
                 mel_scale="htk"):
        if n_mels != 128 or n_stft != 1025 or norm is not None or mel_scale != "htk" or f_min != 0.0:
            raise ValueError("libvfx implements the reference configuration only (128 HTK bands over 1025 bins)")
        self.engine = engine
        self.n_mels, self.sample_rate = n_mels, sample_rate
        self.f_max = f_max if f_max is not None else float(sample_rate // 2)
        self.fb = self.filterbank(n_stft, n_mels, sample_rate, self.f_max)
        engine.set_mel_filterbank(self.fb)

    @staticmethod
    def filterbank(n_freqs, n_mels, sample_rate, f_max):
        """float32 torch-CPU evaluation of the HTK filterbank with the reference's operation
        order (mel_scale.py:131-221), so the table is bit-identical to the reference buffer."""
        hz = torch.linspace(0, sample_rate // 2, n_freqs)
        to_mel = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)
        edges_mel = torch.linspace(to_mel(0.0), to_mel(f_max), n_mels + 2)
        edges = 700.0 * (10.0 ** (edges_mel / 2595.0) - 1.0)
        span = edges[1:] - edges[:-1]
        delta = edges.unsqueeze(0) - hz.unsqueeze(1)
        lower = (-1.0 * delta[:, :-2]) / span[:-1]
        upper = delta[:, 2:] / span[1:]
        return torch.max(torch.zeros(1), torch.min(lower, upper))

    def __call__(self, specgram):
        x = specgram.transpose(-1, -2).contiguous()           # (..., time, freq)
        return self.engine.mel_project(x).transpose(-1, -2)   # (..., n_mels, time)

    forward = __call__


class FDomainHelper:
    """tools/pytorch/modules/fDomainHelper.py:11-113 for subband=None (the only configuration any
    call site uses, fDomainHelper.py:20,25)."""

    def __init__(self, engine, window_size=2048, hop_size=441, center=True, pad_mode="reflect", window="hann",
                 freeze_parameters=True, subband=None):
        if subband is not None:
            raise NotImplementedError("PQMF sub-band analysis is outside the hot path (its filter files are not in the reference)")
        if (window_size, hop_size, pad_mode, window) != (2048, 441, "reflect", "hann") or not center:
            raise ValueError("libvfx implements the reference STFT configuration only (2048/441/hann/reflect/center)")
        self.engine = engine

    def _flat(self, input):
        B, C, L = input.shape
        return input.reshape(B * C, L), (B, C)

    def spectrogram_phase(self, input, eps=0.0):
        """(B, L) -> mag, cos, sin (B, 1, T, 1025); mag = sqrt(clamp(re^2 + im^2, eps, inf)), cos = re / mag,
        sin = im / mag (fDomainHelper.py:60-65; with the reference's default eps = 0 an exactly silent bin gives
        0 / 0 = NaN phases there, as in the reference)."""
        o = self.engine.stft(input, want_mel=False, want_sp=True, want_phase=True, eps=eps)
        return o["sp"][:, None], o["cos"][:, None], o["sin"][:, None]

    def wav_to_spectrogram_phase(self, input, eps=1e-8):
        x, (B, C) = self._flat(input)
        o = self.engine.stft(x, want_mel=False, want_sp=True, want_phase=True, eps=eps)
        shp = (B, C) + tuple(o["sp"].shape[1:])
        return o["sp"].reshape(shp), o["cos"].reshape(shp), o["sin"].reshape(shp)

    def wav_to_spectrogram(self, input, eps=1e-8):
        x, (B, C) = self._flat(input)
        sp = self.engine.stft(x, want_mel=False, want_sp=True, eps=eps)["sp"]
        return sp.reshape((B, C) + tuple(sp.shape[1:]))

    def wav_to_mel(self, input, log10=False):
        """Fused front-end (no (B,C,T,1025) round trip through HBM): (B,C,L) -> (B,C,T,128)."""
        x, (B, C) = self._flat(input)
        mel = self.engine.stft(x, want_mel=True, log10_mel=log10)["mel"]
        return mel.reshape((B, C) + tuple(mel.shape[1:]))

    def istft(self, real, imag, length):
        """(B,1,T,1025) x2 -> (B, length)   [torchlibrosa ISTFT via fDomainHelper.py:30-32]."""
        return self.engine.istft(real[:, 0], imag[:, 0], length)

    def spectrogram_phase_to_wav(self, sps, coss, sins, length):
        B, C = sps.shape[:2]
        flat = lambda t: t.reshape((B * C,) + tuple(t.shape[2:]))
        return self.engine.istft(flat(sps * coss), flat(sps * sins), length).reshape(B, C, length)


class Vocoder:
    """`voicefixer.Vocoder(sample_rate=44100)`: linear mel (B,1,T,128) -> wave (B,1,(T+T%2+4)*441)."""

    def __init__(self, engine, sample_rate=44100):
        assert sample_rate == 44100
        self.engine = engine
        self.rate = sample_rate

    def load_state_dict(self, sd, prefix=""):
        self.engine.load_state_dict(MODEL_VOCODER, fold_weight_norm(sd), prefix)

    def load_from_checkpoint(self, ckpt):
        """The pip package keeps its own generator file beside the Lightning checkpoint (the reference constructs
        `Vocoder(sample_rate=44100)`, models/gsr_voicefixer.py:113, and the weights come from the package cache): a
        `{'generator': state_dict}` container (read_checkpoint), weight-norm pairs folded here, a DataParallel `module.` prefix
        dropped."""
        sd = read_checkpoint(ckpt)
        if sd and all(k.startswith("module.") for k in sd):
            sd = {k[len("module."):]: v for k, v in sd.items()}
        self.load_state_dict(sd)
        return self

    def __call__(self, mel, cuda=None, check=True):
        """`check` = False defers the flag check (one device sync) to the caller: Engine.check_flags, once per file in
        the handlers."""
        assert mel.size()[-1] == 128
        out = self.engine.vocoder(mel[:, 0])
        # only the 16-bit mode can raise a flag here (saturation); the negative-input flag belongs to the UNet stage's prep
        # kernel.  The reference's vocoder has no assert and no sync: precision 0 / 1 pay none either.
        if check and self.engine.precision == 2:
            out = _rerun_if_saturated(self.engine, out, lambda e: e.vocoder(mel[:, 0]))
        return out[:, None]

    forward = __call__


def _rerun_if_saturated(engine, out, call):
    """Every arithmetic mode: a negative value that reached a log10 raises like `to_log`'s assert.  16-bit vocoder
    (precision 2): an activation beyond the fp16 range is clamped by the kernels and reported through a sticky device flag;
    such a call is re-run on the split-bf16 twin of the engine, so the mode is never silently wrong on weights whose
    activations do not fit fp16 (Engine.check_flags: one device sync per call, like `to_log`'s assert)."""
    again = engine.check_flags(call)
    return out if again is None else again


def fold_weight_norm(sd):
    """Replace (weight_g, weight_v) pairs by the plain weight g * v / ||v|| (norm over dims != 0)."""
    out = {}
    for k, v in sd.items():
        if k.endswith("weight_g"):
            base = k[:-len("weight_g")]
            vv = sd[base + "weight_v"].float()
            norm = vv.reshape(vv.shape[0], -1).norm(dim=1).reshape((-1,) + (1,) * (vv.dim() - 1))
            out[base + "weight"] = v.float() * vv / norm
        elif k.endswith("weight_v"):
            continue
        else:
            out[k] = v
    return out


# Lightning checkpoints (`save_hyperparameters()`, models/gsr_voicefixer.py:106) pickle `hyper_parameters` with project
# classes (tools.utils.HParams, ...) that are not importable here, so `torch.load(weights_only=True)` refuses them.
# The fallback below never imports or calls anything the file names: globals on a short allow-list (tensor rebuild
# helpers, storage types, plain containers) resolve normally, EVERY other global becomes an inert stand-in class that
# swallows its constructor arguments and state.  A hostile file can therefore not run code through this reader; what
# it can do is fail to load.
_SAFE_GLOBALS = {
    ("collections", "OrderedDict"), ("collections", "defaultdict"),
    ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"), ("builtins", "set"), ("builtins", "frozenset"),
    ("builtins", "int"), ("builtins", "float"), ("builtins", "bool"), ("builtins", "str"), ("builtins", "bytes"),
    ("builtins", "complex"), ("builtins", "slice"), ("builtins", "range"), ("builtins", "bytearray"),
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
    ("torch._utils", "_rebuild_parameter_with_state"), ("torch._tensor", "_rebuild_from_type_v2"),
    ("torch", "Size"), ("torch", "device"), ("torch", "Tensor"), ("torch.nn.parameter", "Parameter"),
    ("torch.serialization", "_get_layout"), ("_codecs", "encode"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"), ("numpy", "dtype"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"),
}


def _is_safe_global(module, name):
    if (module, name) in _SAFE_GLOBALS:
        return True
    if module == "torch" and (name.endswith("Storage") or name in (
            "float32", "float64", "float16", "bfloat16", "int64", "int32", "int16", "int8", "uint8", "bool")):
        return True
    return False


class _Inert(dict):
    """Stand-in for a class / function the checkpoint names but this reader does not trust or cannot import."""

    def __init__(self, *args, **kwargs):
        dict.__init__(self)

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.update(state)

    def __reduce_ex__(self, protocol):
        return (dict, (dict(self),))

    # list- / set-like reducers (append, extend, add) of stubbed containers
    def append(self, *a):
        pass

    def extend(self, *a):
        pass

    def add(self, *a):
        pass


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if _is_safe_global(module, name):
            return super().find_class(module, name)
        return type(str(name), (_Inert,), {"__module__": "voicefixer_main_amd.models"})


class _PickleModule:
    Unpickler = _Unpickler
    load = staticmethod(lambda f, **kw: _Unpickler(f, **kw).load())
    __name__ = "pickle"


def read_checkpoint(path):
    """Lightning .ckpt (``{'state_dict': ..., 'hyper_parameters': ...}``, eval_gsr_voicefixer.py:33), a GAN-style container
    (``{'generator': state_dict, ...}``: the pip vocoder's own file; ``{'model': ...}``) or a bare state_dict file ->
    state_dict of CPU tensors.  Tries torch's `weights_only` reader first; a file that only fails there because it pickles
    project classes is re-read with the allow-listing unpickler above (nothing the file names is imported or executed)."""
    try:
        obj = torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError:
        obj = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_PickleModule)
    sd = obj
    for key in ("state_dict", "generator", "model"):
        if isinstance(obj, dict) and isinstance(obj.get(key), dict) and obj[key] and \
                all(isinstance(v, torch.Tensor) for v in obj[key].values()):
            sd = obj[key]
            break
    if not isinstance(sd, dict) or not sd or not all(isinstance(v, torch.Tensor) for v in sd.values()):
        raise ValueError("%s holds no state_dict of tensors" % path)
    return sd


class _Base:
    def __init__(self, hp=None, channels=1, type_target="vocals", device="cuda:0", engine=None):
        self.hp = hp
        self.channels = channels
        self.type_target = type_target
        self.engine = engine if engine is not None else Engine(device)
        self.device = self.engine.device
        self.sampling_rate = _hp_get(hp, "data", "sampling_rate", default=44100)
        self.f_helper = FDomainHelper(
            self.engine,
            window_size=_hp_get(hp, "model", "window_size", default=2048),
            hop_size=_hp_get(hp, "model", "hop_size", default=441),
            center=True,
            pad_mode=_hp_get(hp, "model", "pad_mode", default="reflect"),
            window=_hp_get(hp, "model", "window", default="hann"))
        self.mel = MelScale(self.engine, n_mels=_hp_get(hp, "model", "mel_freq_bins", default=128),
                            sample_rate=self.sampling_rate,
                            n_stft=_hp_get(hp, "model", "window_size", default=2048) // 2 + 1)
        self.vocoder = Vocoder(self.engine, sample_rate=44100)
        self.training = False

    # nn.Module-ish surface the handlers touch
    def eval(self):
        self.training = False
        return self

    def to(self, device):
        if torch.device(device).type != "cuda":
            raise RuntimeError("this model only runs on the MI355X (requested %s)" % device)
        return self

    def get_vocoder(self):
        return self.vocoder

    def get_f_helper(self):
        return self.f_helper

    def pre(self, input):
        sp, _, _ = self.f_helper.wav_to_spectrogram_phase(input)
        mel_orig = self.mel(sp.permute(0, 1, 3, 2)).permute(0, 1, 3, 2)
        return sp, mel_orig

    def load_from_checkpoint(self, ckpt):
        self.load_state_dict(read_checkpoint(ckpt))
        return self


class VoiceFixer(_Base):
    """models/gsr_voicefixer.py:93-193 (inference surface): mel ResUNet + TFGAN vocoder."""

    unet_prefix = "generator.analysis_module."

    def load_state_dict(self, sd, strict=True):
        keys = list(sd.keys())
        if any(k.startswith(self.unet_prefix) for k in keys):
            self.engine.load_state_dict(MODEL_UNET_MEL, sd, self.unet_prefix)
        elif any(k.startswith("encoder_block1.") for k in keys):
            self.engine.load_state_dict(MODEL_UNET_MEL, sd)
        elif strict:
            raise KeyError("no ResUNet weights ('%s*') in the state_dict" % self.unet_prefix)
        for pfx in ("vocoder.model.", "vocoder."):
            sub = {k[len(pfx):]: v for k, v in sd.items() if k.startswith(pfx)}
            if any(k.startswith("condnet.") for k in sub):
                self.vocoder.load_state_dict(sub)
                break
        if "mel.fb" in sd:
            self.engine.set_mel_filterbank(sd["mel.fb"])
        return self

    def forward(self, mel_orig, check=True):
        """mel_orig (B,1,T,128) linear, non-negative -> {'mel': log10 estimate}  (Generator.forward,
        gsr_voicefixer.py:86-91).  `check` reproduces to_log's assert (one device sync)."""
        out = self.engine.resunet_mel(mel_orig[:, 0])[:, None]
        if check:
            # to_log's assert; a saturation bit left by a deferred vocoder check (Vocoder.__call__(check=False), raw engine calls)
            # is NOT consumed here -- it stays raised for that check
            self.engine.check_negative_input()
        return {"mel": out}

    __call__ = forward

    def restore(self, wav, unify_energy=False):
        """Fused handler() segment body: wav (B,1,L) or (B,L) -> restored wav of the same shape."""
        squeeze = wav.dim() == 3
        x = wav[:, 0] if squeeze else wav
        out = self.engine.restore_gsr(x, unify_energy=unify_energy)
        out = _rerun_if_saturated(self.engine, out, lambda e: e.restore_gsr(x, unify_energy=unify_energy))
        return out[:, None] if squeeze else out

    def restore_list(self, wavs, unify_energy=False, max_batch=128):
        """A test set of clips of ARBITRARY lengths (what the reference's harness iterates, one handler call per file:
        evaluation_proc/eval.py:119-134): list of 1-D tensors -> list of restored 1-D tensors in the same order.  The clips go
        through the library sorted by length, up to `max_batch` per call, as padded batches with their lengths
        (vfx_restore_gsr_varlen: every clip's result is the one its own batch-of-one call gives; inside a call the mel ResUNet
        runs once per padded frame count over all its clips, the vocoder per run of clips of similar length); with
        torch.distributed initialised the list is dealt over the ranks by length and gathered on rank 0
        (dist.restore_sharded_lengths).  The 16-bit mode's re-run guarantee holds per batch."""
        from . import dist as vdist
        fn = vdist.checked_restore(self.engine, unify_energy=unify_energy)
        return vdist.restore_sharded_lengths(fn, wavs, self.device, max_batch=max_batch)


class SSR_UNet(_Base):
    """models/ssr_unet.py:56-155 / models/gsr_unet.py (identical inference surface): the
    spectrogram-domain ResUNet of models/components/unet_v2.py."""

    unet_prefix = "generator.unet."

    def load_state_dict(self, sd, strict=True):
        keys = list(sd.keys())
        if any(k.startswith(self.unet_prefix) for k in keys):
            sub = {k: v for k, v in sd.items() if ".f_helper." not in k}
            self.engine.load_state_dict(MODEL_UNET_SPEC, sub, self.unet_prefix)
        elif any(k.startswith("encoder_block1.") for k in keys):
            self.engine.load_state_dict(MODEL_UNET_SPEC, {k: v for k, v in sd.items() if not k.startswith("f_helper.")})
        elif strict:
            raise KeyError("no ResUNet weights ('%s*') in the state_dict" % self.unet_prefix)
        return self

    def forward(self, sp, wav):
        """sp (B,1,T,1025) magnitude, wav (B,1,L) -> {'wav': (B,1,L), 'clean': sp}."""
        out = self.engine.resunet_spec(sp[:, 0], wav[:, 0])
        return {"wav": out[:, None], "clean": sp}

    __call__ = forward

    def restore_list(self, wavs, max_batch=16):
        """A test set of clips of ARBITRARY lengths through `pre` + `forward` (eval_ssr_unet.py:77-114, one handler call per file
        in the reference): list of 1-D tensors -> list of restored 1-D tensors in the same order.  Clips whose frame counts pad
        to the same multiple of 64 share one call of the library as a padded batch with their lengths (vfx_restore_ssr_varlen);
        with torch.distributed initialised the list is dealt over the ranks (dist.restore_sharded_lengths)."""
        from . import dist as vdist
        eng = self.engine

        def fn(x, lengths=None):
            if lengths is None:
                return eng.resunet_spec(eng.stft(x, want_mel=False, want_sp=True)["sp"], x)
            return eng.restore_ssr_varlen(x, lengths)
        fn.bucket_key = eng.padded_frames
        fn.bucket_len = lambda L: eng.padded_frames(L) * eng.hop - 1      # one plan per (B, bucket), cf. dist.checked_restore
        return vdist.restore_sharded_lengths(fn, wavs, self.device, max_batch=max_batch)


GSR_UNet = SSR_UNet
