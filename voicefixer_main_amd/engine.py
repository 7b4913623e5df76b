"""Thin Python wrapper over one libvfx handle: torch tensors in, torch tensors out.

All compute happens in the HIP library; this file only allocates outputs with torch,
passes raw device pointers + the current HIP stream, and raises on any failure.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import MODEL_FRONTEND, MODEL_UNET_MEL, MODEL_UNET_SPEC, MODEL_VOCODER

N_BINS = 1025
N_MELS = 128


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _dev_f32(t, device):
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(np.asarray(t))
    return t.to(device=device, dtype=torch.float32).contiguous()


class Engine:
    """One libvfx handle on one GPU (not thread-safe, like the reference's module-level model)."""

    def __init__(self, device="cuda:0", config=None):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("voicefixer_main_amd runs on an MI355X GPU only (got device %s); "
                               "there is no CPU path" % device)
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible to PyTorch")
        torch.cuda.init()
        self.cfg = _lib.VfxConfig()
        self.lib.vfx_default_config(ctypes.byref(self.cfg))
        if config:
            for k, v in config.items():
                cur = getattr(self.cfg, k)
                if hasattr(cur, "__len__"):
                    for i, x in enumerate(v):
                        cur[i] = x
                else:
                    setattr(self.cfg, k, v)
        h = ctypes.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(self.lib.vfx_create(idx, ctypes.byref(self.cfg), ctypes.byref(h)), "vfx_create")
        self.h = h
        self.loaded = set()
        self._state = {}        # model id -> (state_dict, prefix) as loaded, for the stricter-arithmetic twin
        self._strict = None

    def close(self):
        if getattr(self, "_strict", None) is not None:
            self._strict.close()
            self._strict = None
        if getattr(self, "h", None):
            self.lib.vfx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @property
    def hop(self):
        return self.cfg.hop

    def frames(self, L):
        return L // self.cfg.hop + 1

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, model, state_dict, prefix=""):
        """Upload a (reference-keyed) state_dict for `model` and finalize it."""
        n = 0
        for k, v in state_dict.items():
            if prefix:
                if not k.startswith(prefix):
                    continue
                k = k[len(prefix):]
            if k.endswith("num_batches_tracked"):
                continue
            a = np.ascontiguousarray(v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v),
                                     dtype=np.float32)
            shape = (ctypes.c_int64 * max(a.ndim, 1))(*a.shape)
            _lib.check(self.lib.vfx_load_tensor(self.h, model, k.encode(), a.ctypes.data_as(ctypes.c_void_p), shape, a.ndim),
                       "vfx_load_tensor(%s)" % k)
            n += 1
        if n == 0:
            raise RuntimeError("no tensors with prefix %r in the state_dict" % prefix)
        _lib.check(self.lib.vfx_finalize_weights(self.h, model), "vfx_finalize_weights")
        self.loaded.add(model)
        self._state[model] = (state_dict, prefix)
        if self._strict is not None:
            self._strict.load_state_dict(model, state_dict, prefix)

    @property
    def precision(self):
        return int(self.cfg.precision)

    def strict_twin(self):
        """A second handle on the same device with the same weights in split-bf16 arithmetic (precision 1): what a call
        is re-run on when the 16-bit vocoder reports a clamped activation (VFX_FLAG_F16_SATURATED)."""
        if self._strict is None:
            cfg = {f[0]: getattr(self.cfg, f[0]) for f in self.cfg._fields_}
            cfg = {k: (list(v) if hasattr(v, "__len__") else v) for k, v in cfg.items()}
            cfg["precision"] = 1
            twin = Engine(self.device, config=cfg)
            for model, (sd, prefix) in self._state.items():
                twin.load_state_dict(model, sd, prefix)
            self._strict = twin
        return self._strict

    def set_mel_filterbank(self, fb):
        self.load_state_dict(MODEL_FRONTEND, {"mel.fb": fb})

    def reserve(self, model, B, T):
        _lib.check(self.lib.vfx_reserve(self.h, model, B, T), "vfx_reserve")

    def unpin_plans(self):
        """Every hipGraph captured from this handle has been destroyed: its plans may be evicted and the arena may grow again
        (a handle with a captured plan refuses to move its arena: reserve() the largest shape before capturing)."""
        _lib.check(self.lib.vfx_unpin_plans(self.h), "vfx_unpin_plans")

    def replay(self, graph):
        """Replay a torch.cuda.CUDAGraph captured from this engine's calls on the current stream, inside the device's turn:
        captured calls are exempt from the library's one-stream-at-a-time rule (vfx.h, "Turns"), their replay is not a call of
        the library -- this brackets it so that it cannot overlap a live call on another stream of the device."""
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        s = self._stream()
        _lib.check(self.lib.vfx_turn_begin(idx, s), "vfx_turn_begin")
        try:
            graph.replay()
        finally:
            _lib.check(self.lib.vfx_turn_end(idx, s), "vfx_turn_end")

    def workspace_bytes(self, model, B, T):
        return int(self.lib.vfx_workspace_bytes(self.h, model, B, T))

    def take_flags(self, mask=None):
        """Read and clear the sticky device flags (one device sync); with `mask`, only those bits -- the others stay raised."""
        f = ctypes.c_int(0)
        if mask is None:
            _lib.check(self.lib.vfx_take_flags(self.h, self._stream(), ctypes.byref(f)), "vfx_take_flags")
        else:
            _lib.check(self.lib.vfx_take_flags_masked(self.h, self._stream(), int(mask), ctypes.byref(f)), "vfx_take_flags_masked")
        return f.value

    # ------------------------------------------------------------------ stages
    def stft(self, wav, want_mel=True, want_sp=False, want_phase=False, log10_mel=False, eps=1e-8):
        """wav (B, L) -> dict with any of mel (B,T,128), sp / cos / sin (B,T,1025).  `eps` is the clamp on the power
        (fDomainHelper.py:60-65); the fused mel output exists for the handlers' eps = 1e-8 only."""
        wav = _dev_f32(wav, self.device)
        B, L = wav.shape
        T = self.frames(L)
        eps = float(eps)
        if eps < 0.0:
            raise ValueError("eps must be >= 0")
        out = {}
        if want_mel:
            if eps != 1e-8:
                raise ValueError("the fused mel output is defined for eps = 1e-8 (wav_to_spectrogram_phase) only")
            out["mel"] = torch.empty((B, T, N_MELS), device=self.device, dtype=torch.float32)
        if want_sp:
            out["sp"] = torch.empty((B, T, N_BINS), device=self.device, dtype=torch.float32)
        if want_phase:
            out["cos"] = torch.empty((B, T, N_BINS), device=self.device, dtype=torch.float32)
            out["sin"] = torch.empty((B, T, N_BINS), device=self.device, dtype=torch.float32)
        if eps == 1e-8:
            _lib.check(self.lib.vfx_stft_mel(self.h, _ptr(wav), B, L, _ptr(out.get("mel")), _ptr(out.get("sp")),
                                             _ptr(out.get("cos")), _ptr(out.get("sin")), int(log10_mel), self._stream()),
                       "vfx_stft_mel")
        else:
            _lib.check(self.lib.vfx_stft_phase(self.h, _ptr(wav), B, L, _ptr(out.get("sp")), _ptr(out.get("cos")),
                                               _ptr(out.get("sin")), eps, self._stream()), "vfx_stft_phase")
        return out

    def mel_project(self, sp):
        """sp (..., 1025) -> (..., 128)."""
        sp = _dev_f32(sp, self.device)
        rows = sp.numel() // N_BINS
        mel = torch.empty(sp.shape[:-1] + (N_MELS,), device=self.device, dtype=torch.float32)
        _lib.check(self.lib.vfx_mel_project(self.h, _ptr(sp), rows, _ptr(mel), self._stream()), "vfx_mel_project")
        return mel

    def istft(self, re, im, length):
        re, im = _dev_f32(re, self.device), _dev_f32(im, self.device)
        B, T, _ = re.shape
        wav = torch.empty((B, length), device=self.device, dtype=torch.float32)
        _lib.check(self.lib.vfx_istft(self.h, _ptr(re), _ptr(im), B, T, length, _ptr(wav), self._stream()), "vfx_istft")
        return wav

    def spectral_metrics(self, est, target):
        """Per-clip (LSD, SiSpec dB) of est vs target, (B, T, F) or (B, 1, T, F) each -> (B, 2)."""
        est, target = _dev_f32(est, self.device), _dev_f32(target, self.device)
        assert est.shape == target.shape
        F = est.shape[-1]
        T = est.shape[-2]
        B = est.numel() // (T * F)
        out = torch.empty((B, 2), device=self.device, dtype=torch.float32)
        _lib.check(self.lib.vfx_spectral_metrics(self.h, _ptr(est), _ptr(target), B, T, F, _ptr(out), self._stream()),
                   "vfx_spectral_metrics")
        return out

    def chunk_gather(self, x, win, hop, lead, n_chunks):
        """F.unfold with zero padding: x (B, L) -> (B, n_chunks, win), chunk k = x[k*hop - lead : ... + win]."""
        x = _dev_f32(x, self.device)
        B, L = x.shape
        chunks = torch.empty((B, n_chunks, win), device=self.device, dtype=torch.float32)
        _lib.check(self.lib.vfx_chunk_gather(self.h, _ptr(x), B, L, win, hop, lead, n_chunks, _ptr(chunks),
                                             self._stream()), "vfx_chunk_gather")
        return chunks

    def chunk_ola(self, frames, window, scale, hop, lead, length):
        """Synthesis window (or scale) + F.fold: frames (B, n_chunks, win) -> (B, length)."""
        frames = _dev_f32(frames, self.device)
        B, n_chunks, win = frames.shape
        if window is not None:
            window = _dev_f32(window, self.device)
            assert window.numel() == win
        y = torch.empty((B, length), device=self.device, dtype=torch.float32)
        _lib.check(self.lib.vfx_chunk_ola(self.h, _ptr(frames), _ptr(window) if window is not None else None,
                                          float(scale), B, n_chunks, win, hop, lead, length, _ptr(y),
                                          self._stream()), "vfx_chunk_ola")
        return y

    def resunet_mel(self, mel_linear):
        """Generator.forward: linear mel (B,T,128) -> log10 mel (B,T,128)."""
        mel = _dev_f32(mel_linear, self.device)
        B, T, _ = mel.shape
        out = torch.empty_like(mel)
        _lib.check(self.lib.vfx_resunet_mel(self.h, _ptr(mel), B, T, _ptr(out), self._stream()), "vfx_resunet_mel")
        return out

    def resunet_spec(self, sp, wav):
        sp, wav = _dev_f32(sp, self.device), _dev_f32(wav, self.device)
        B, T, _ = sp.shape
        L = wav.shape[-1]
        out = torch.empty((B, L), device=self.device, dtype=torch.float32)
        _lib.check(self.lib.vfx_resunet_spec(self.h, _ptr(sp), _ptr(wav), B, T, L, _ptr(out), self._stream()),
                   "vfx_resunet_spec")
        return out

    def vocoder_out_len(self, T):
        return int(self.lib.vfx_vocoder_out_len(self.h, T))

    def vocoder(self, mel_linear):
        mel = _dev_f32(mel_linear, self.device)
        B, T, _ = mel.shape
        out = torch.empty((B, self.vocoder_out_len(T)), device=self.device, dtype=torch.float32)
        _lib.check(self.lib.vfx_vocoder(self.h, _ptr(mel), B, T, _ptr(out), self._stream()), "vfx_vocoder")
        return out

    def restore_gsr(self, wav, unify_energy=False, want_logmel=False, out=None):
        """Whole handler() segment body for a batch: wav (B, L) -> restored (B, L)."""
        wav = _dev_f32(wav, self.device)
        B, L = wav.shape
        if out is None:
            out = torch.empty_like(wav)
        logmel = torch.empty((B, self.frames(L), N_MELS), device=self.device, dtype=torch.float32) if want_logmel else None
        _lib.check(self.lib.vfx_restore_gsr(self.h, _ptr(wav), B, L, _ptr(out), _ptr(logmel), int(bool(unify_energy)),
                                            self._stream()), "vfx_restore_gsr")
        return (out, logmel) if want_logmel else out

    def padded_frames(self, L):
        """The ResUNet's padded frame count of a clip of L samples (unet.py:75-77): 64 * ceil(T / 64) -- the bucket key of
        restore_ssr_varlen, and the granule restore_gsr_varlen's callers pad a batch's row length to (one cached plan per row length)."""
        return -(-self.frames(L) // 64) * 64

    def supports_varlen(self):
        """False for the one configuration vfx_restore_gsr_varlen refuses: the 16-bit mode on the fp32 trunk (VFX_TUNE_F32_TRUNK, or
        a ResStack slope outside (0, 1]) -- the persistent C = 64 kernel has no register left for a clip's length there.  Callers
        that bucket clips (dist.checked_restore, VoiceFixer.restore_list) then batch EQUAL lengths only, as rounds 1-4 did."""
        if self.precision != 2:
            return True
        slope = float(self.cfg.voc_res_slope)
        return not (int(self.cfg.tuning) & _lib.TUNE_F32_TRUNK) and 0.0 < slope <= 1.0

    def restore_gsr_varlen(self, wav, lengths, unify_energy=False, want_logmel=False, out=None):
        """The handler() segment body for a batch of clips of UNEQUAL length: wav (B, Lmax), clip b = wav[b, :lengths[b]]
        -> restored (B, Lmax), zero past a clip's end.  Every clip gets what its own restore_gsr(wav[b:b+1, :lengths[b]])
        computes (vfx_restore_gsr_varlen).  Any mix of lengths (round 6): the library runs the mel ResUNet once per padded frame
        count among the clips and the vocoder once over the whole batch."""
        wav = _dev_f32(wav, self.device)
        B, L = wav.shape
        lengths = [int(v) for v in lengths]
        if len(lengths) != B:
            raise ValueError("restore_gsr_varlen: %d lengths for %d clips" % (len(lengths), B))
        arr = (ctypes.c_int * B)(*lengths)
        if out is None:
            out = torch.empty_like(wav)
        logmel = torch.empty((B, self.frames(L), N_MELS), device=self.device, dtype=torch.float32) if want_logmel else None
        _lib.check(self.lib.vfx_restore_gsr_varlen(self.h, _ptr(wav), B, L, arr, _ptr(out), _ptr(logmel),
                                                   int(bool(unify_energy)), self._stream()), "vfx_restore_gsr_varlen")
        return (out, logmel) if want_logmel else out

    def restore_ssr_varlen(self, wav, lengths, out=None):
        """ssr_unet / gsr_unet forward (sp = |STFT(wav)|, model(sp, wav)) for a batch of clips of UNEQUAL length: wav (B, Lmax),
        clip b = wav[b, :lengths[b]] -> (B, Lmax), zero past a clip's end; one `padded_frames` bucket per call."""
        wav = _dev_f32(wav, self.device)
        B, L = wav.shape
        lengths = [int(v) for v in lengths]
        if len(lengths) != B:
            raise ValueError("restore_ssr_varlen: %d lengths for %d clips" % (len(lengths), B))
        arr = (ctypes.c_int * B)(*lengths)
        if out is None:
            out = torch.empty_like(wav)
        _lib.check(self.lib.vfx_restore_ssr_varlen(self.h, _ptr(wav), B, L, arr, _ptr(out), self._stream()), "vfx_restore_ssr_varlen")
        return out

    def check_negative_input(self):
        """`to_log`'s assert alone (pytorch_util.py:158): reads and clears ONLY the negative-input bit -- a saturation bit a
        deferred vocoder check still has to see stays raised."""
        if self.take_flags(_lib.FLAG_NEGATIVE_INPUT) & _lib.FLAG_NEGATIVE_INPUT:
            raise AssertionError("to_log: input has negative values")

    def check_flags(self, rerun=None):
        """Reads and clears the handle's sticky device flags (one device sync, like `to_log`'s assert in the reference).
        A negative value reached a log10 -> AssertionError (pytorch_util.py:158), in every arithmetic mode.  16-bit vocoder
        (precision 2): an activation beyond the fp16 range was clamped -> the call is repeated on the split-bf16 twin
        (`rerun(twin_engine)`), or RuntimeError when no `rerun` is given.  Returns rerun's result, or None."""
        flags = self.take_flags()
        if flags & _lib.FLAG_NEGATIVE_INPUT:
            raise AssertionError("to_log: input has negative values")
        if flags & _lib.FLAG_F16_SATURATED:
            if rerun is None:
                raise RuntimeError("16-bit vocoder: an activation left the fp16 range (VFX_FLAG_F16_SATURATED); "
                                   "re-run with precision 1")
            import warnings
            warnings.warn("16-bit vocoder: an activation left the fp16 range; this call is re-run with split-bf16 operands")
            try:
                twin = self.strict_twin()
            except RuntimeError as e:      # a second copy of the weights + its arena on the same GPU, mid-run
                raise RuntimeError("16-bit vocoder: the split-bf16 twin this call must be re-run on could not be created "
                                   "(%s); call Engine.strict_twin() once at start-up to reserve it, or use precision 1" % e)
            return rerun(twin)
        return None

    def restore_gsr_checked(self, wav, unify_energy=False, out=None):
        """restore_gsr + check_flags: never silently wrong in the 16-bit mode, whoever the caller is (models, dist, bench)."""
        res = self.restore_gsr(wav, unify_energy=unify_energy, out=out)
        again = self.check_flags(lambda e: e.restore_gsr(wav, unify_energy=unify_energy, out=out))
        return res if again is None else again

    def restore_gsr_varlen_checked(self, wav, lengths, unify_energy=False, out=None):
        """restore_gsr_varlen + check_flags (cf. restore_gsr_checked)."""
        res = self.restore_gsr_varlen(wav, lengths, unify_energy=unify_energy, out=out)
        again = self.check_flags(lambda e: e.restore_gsr_varlen(wav, lengths, unify_energy=unify_energy, out=out))
        return res if again is None else again

    def vocoder_checked(self, mel_linear):
        res = self.vocoder(mel_linear)
        again = self.check_flags(lambda e: e.vocoder(mel_linear))
        return res if again is None else again

    def profile_begin(self):
        _lib.check(self.lib.vfx_profile_begin(self.h), "vfx_profile_begin")

    def profile_end(self):
        """-> (launches, total_ms, total_flops) of the tap-convolution launches since profile_begin."""
        n, ms, fl = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
        _lib.check(self.lib.vfx_profile_end(self.h, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl)), "vfx_profile_end")
        return n.value, ms.value, fl.value

    # ------------------------------------------------------------------ kernel-level ops (tests: libvfx_test.so, include/vfx_test.h)
    def op_conv(self, x, weight, scale=None, shift=None, act=0, slope=0.0, bias=None, residual=None, dil_w=1,
                reflect_w=False):
        """x (B,H,W,Cin) channels-last; weight (Cout,Cin,kh,kw) torch layout (host)."""
        x = _dev_f32(x, self.device)
        B, H, W, Cin = x.shape
        w = np.ascontiguousarray(np.asarray(weight, dtype=np.float32))
        Cout, _, kh, kw = w.shape
        y = torch.empty((B, H, W, Cout), device=self.device, dtype=torch.float32)
        hp = lambda a: None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))
        scale, shift, bias = hp(scale), hp(shift), hp(bias)
        cp = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
        res = None if residual is None else _dev_f32(residual, self.device)
        _lib.check(_lib.load_test().vfx_op_conv(self.h, _ptr(x), B, H, W, Cin, cp(w), Cout, kh, kw, dil_w, int(reflect_w), cp(scale),
                                        cp(shift), act, float(slope), cp(bias), _ptr(res), _ptr(y), self._stream()),
                   "vfx_op_conv")
        return y

    def op_resblock(self, x, w1, b1, w2, b2, dil, slope=0.01, fused=True):
        """One ResStack layer on x (B, T, C) channels-last; w1 / w2 (C, C, 3), b1 / b2 (C) torch layout (host)."""
        x = _dev_f32(x, self.device)
        B, T, C = x.shape
        hp = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float32))
        w1, b1, w2, b2 = hp(w1), hp(b1), hp(w2), hp(b2)
        cp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        y = torch.empty_like(x)
        _lib.check(_lib.load_test().vfx_op_resblock(self.h, _ptr(x), B, T, C, cp(w1), cp(b1), cp(w2), cp(b2), int(dil), float(slope),
                                            int(bool(fused)), _ptr(y), self._stream()), "vfx_op_resblock")
        return y

    def op_resblock_pair(self, x, layer_a, dil_a, layer_b, dil_b, slope=0.01):
        """Two consecutive ResStack layers as one launch (16-bit mode, C = 64); layer_* = (w1, b1, w2, b2) in torch layout."""
        x = _dev_f32(x, self.device)
        B, T, C = x.shape
        hp = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float32))
        la, lb = [hp(a) for a in layer_a], [hp(a) for a in layer_b]
        cp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        y = torch.empty_like(x)
        _lib.check(_lib.load_test().vfx_op_resblock_pair(self.h, _ptr(x), B, T, C, cp(la[0]), cp(la[1]), cp(la[2]), cp(la[3]), int(dil_a),
                                                 cp(lb[0]), cp(lb[1]), cp(lb[2]), cp(lb[3]), int(dil_b), float(slope), _ptr(y),
                                                 self._stream()), "vfx_op_resblock_pair")
        return y

    def op_block2d(self, x, w1, sc1, sh1, w2, sc2, sh2, slope=0.01):
        """One fused ConvBlockRes (identity shortcut) on x (B, H, W, C) channels-last; w1 / w2 (C, C, 3, 3) and the folded
        BatchNorm affines (C) in torch layout on the host."""
        x = _dev_f32(x, self.device)
        B, H, W, C = x.shape
        hp = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float32))
        w1, sc1, sh1, w2, sc2, sh2 = hp(w1), hp(sc1), hp(sh1), hp(w2), hp(sc2), hp(sh2)
        cp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        y = torch.empty_like(x)
        _lib.check(_lib.load_test().vfx_op_block2d(self.h, _ptr(x), B, H, W, C, cp(w1), cp(sc1), cp(sh1), cp(w2), cp(sc2), cp(sh2),
                                           float(slope), _ptr(y), self._stream()), "vfx_op_block2d")
        return y

    def op_conv_transpose(self, x, weight, stride, prune_w=False, scale=None, shift=None, act=0, slope=0.0, bias=None):
        x = _dev_f32(x, self.device)
        B, H, W, Cin = x.shape
        w = np.ascontiguousarray(np.asarray(weight, dtype=np.float32))
        _, Cout, kh, kw = w.shape
        if kh == 3:
            shape = (B, 2 * H, 2 * W if prune_w else 2 * W + 1, Cout)
        else:
            shape = (B, 1, W * stride, Cout)
        y = torch.empty(shape, device=self.device, dtype=torch.float32)
        hp = lambda a: None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))
        scale, shift, bias = hp(scale), hp(shift), hp(bias)
        cp = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
        _lib.check(_lib.load_test().vfx_op_conv_transpose(self.h, _ptr(x), B, H, W, Cin, cp(w), Cout, kh, kw, stride, int(prune_w),
                                                  cp(scale), cp(shift), act, float(slope), cp(bias), _ptr(y), self._stream()),
                   "vfx_op_conv_transpose")
        return y
