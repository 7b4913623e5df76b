"""`handler()` plugins for the reference's `evaluation()` harness, backed by libvfx.

The harness contract (evaluation_proc/eval.py:119-134): ``handler(input, output, target, ckpt,
device, needrefresh, meta) -> dict`` once per wav file; the returned dict is JSON-dumped.  These
mirror eval_gsr_voicefixer.py:37-77 and eval_ssr_unet.py:95-143 line by line (60-s segments, peak
normalisation, trim_center, concatenation), with the models of `voicefixer_main_amd.models`.

Wav I/O uses the stdlib `wave` module + numpy (the reference needs librosa / soundfile, which
are outside the hot path and not installed here): PCM16 mono/stereo in, PCM16 out, polyphase
resampling (scipy) when the file's rate is not 44.1 kHz.
"""
import wave

import numpy as np
import torch

from . import models
from .models import from_log, tensor2numpy, to_log

EPS = 1e-12        # evaluation_proc/metrics.py:16
EPS_UNIFY = 1e-8   # evaluation_proc/utils.py:8 (energy_unify)
SEG_SECONDS = 60
MAX_SEGMENT_BATCH = 4   # full segments of one file per call when no per-segment metrics are asked for (4 x 60 s: 3 GB of workspace)


# ----------------------------------------------------------------------------------------
# wav I/O (host side of the boundary)
# ----------------------------------------------------------------------------------------
def load_wav(path, sample_rate=44100):
    """Mono float32 in [-1, 1] at `sample_rate` (librosa.load(path, sr) semantics, tools/utils.py:46-48)."""
    with wave.open(path, "rb") as f:
        n, ch, width, sr = f.getnframes(), f.getnchannels(), f.getsampwidth(), f.getframerate()
        raw = f.readframes(n)
    if width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError("unsupported sample width %d in %s" % (width, path))
    x = x.reshape(-1, ch).mean(axis=1)
    if sr != sample_rate:
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(sr, sample_rate)
        x = resample_poly(x, sample_rate // g, sr // g).astype(np.float32)
    return x


def save_wave(frames, fname, sample_rate=44100):
    """tools/file/wav.py:10-27: float frames in [-1, 1] -> PCM16 file."""
    x = np.asarray(frames)
    if x.ndim > 1:
        x = x.reshape(-1, x.shape[-1])[0] if x.shape[-1] > 8 else x[..., 0]
    pcm = (x.astype(np.float64) * 2 ** 15).astype(np.short)
    with wave.open(fname, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(sample_rate)
        f.writeframes(pcm.tobytes())


def trim_center(est, ref):
    """tools/utils.py:57-70 (including the empty slice when the length difference is 1)."""
    diff = abs(est.shape[-1] - ref.shape[-1])
    if est.shape[-1] == ref.shape[-1]:
        return est, ref
    min_len = min(est.shape[-1], ref.shape[-1])
    h = int(diff // 2)
    if est.shape[-1] > ref.shape[-1]:
        est = est[..., h:-h]
    else:
        ref = ref[..., h:-h]
    return est[..., :min_len], ref[..., :min_len]


def amp_to_original_f(mel_sp_est, mel_sp_target, cutoff=0.2):
    """tools/utils.py:50-55."""
    hi = int(mel_sp_target.size()[-1] * cutoff)
    e_est = torch.mean(mel_sp_est[..., 5:hi], dim=(2, 3))
    e_tgt = torch.mean(mel_sp_target[..., 5:hi], dim=(2, 3))
    return mel_sp_est * (e_tgt / e_est)[..., None, None], mel_sp_target


# ----------------------------------------------------------------------------------------
# spectral metrics (evaluation_proc/metrics.py:83-95, utils.py:81-101).  `lsd` / `sispec` are the formulas in torch
# (any device); the handlers call `device_metrics`, which runs them in libvfx (vfx_spectral_metrics: one pass over the
# data, no (B,T,F) temporaries) and returns Python floats with one device-to-host copy per pair.
# ----------------------------------------------------------------------------------------
def device_metrics(engine, est, target):
    """-> (lsd, sispec_db) of a (B, 1, T, F) pair as the reference reports them (batch means)."""
    m = engine.spectral_metrics(est, target).mean(dim=0).tolist()
    return m[0], m[1]


def _pow_p_norm(x):
    return torch.pow(torch.norm(x.reshape(x.shape[0], -1), p=2, dim=1), 2).reshape((-1,) + (1,) * (x.dim() - 1))


def lsd(est, target):
    v = torch.log10((target ** 2 / ((est + EPS) ** 2)) + EPS) ** 2
    return torch.mean(torch.mean(v, dim=3) ** 0.5, dim=2)[..., None, None]


def sispec(est, target):
    scale = torch.sum(est * target, dim=tuple(range(2, est.dim())), keepdim=True)
    tgt = scale * target / (_pow_p_norm(target) + EPS_UNIFY)
    noise = est - tgt
    loss = 10 * torch.log10(_pow_p_norm(tgt) / (_pow_p_norm(noise) + EPS) + EPS)
    return torch.sum(loss) / loss.size()[0]


def ssim(est, target, win_size=7, data_range=2.0, k1=0.01, k2=0.03):
    """evaluation_proc/metrics.py:97-106: per (batch, channel) `skimage.metrics.structural_similarity(est, target,
    win_size=7)` -> (B, C, 1, 1).  skimage's defaults restated on the device: uniform 7x7 window, sample covariance
    (normalised by N-1), K1 = 0.01, K2 = 0.03, the mean taken over the region whose window lies inside the image
    (the (win_size-1)//2 border is cropped), and -- for floating-point images without an explicit data_range, which
    is how the reference calls it -- data_range = 2 (the dtype range [-1, 1] of the skimage releases that accept that
    call).  The reference moves both spectrograms to the host for this; here they stay on the device."""
    x, y = est.double(), target.double()
    B, C, H, W = x.shape
    x, y = x.reshape(B * C, 1, H, W), y.reshape(B * C, 1, H, W)
    box = lambda t: torch.nn.functional.avg_pool2d(t, win_size, stride=1)     # 'valid' box mean == uniform_filter, cropped
    n = float(win_size * win_size)
    cov_norm = n / (n - 1.0)
    ux, uy = box(x), box(y)
    vx = cov_norm * (box(x * x) - ux * ux)
    vy = cov_norm * (box(y * y) - uy * uy)
    vxy = cov_norm * (box(x * y) - ux * uy)
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    s_map = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
    return s_map.mean(dim=(1, 2, 3)).reshape(B, C, 1, 1)


# ----------------------------------------------------------------------------------------
# eval_gsr_voicefixer.py
# ----------------------------------------------------------------------------------------
_state = {"model": None, "hp": None}


def set_hparams(hp):
    _state["hp"] = hp


def refresh_model(ckpt, cls):
    model = cls(_state["hp"], channels=2, type_target="vocals")
    if isinstance(ckpt, dict):
        model.load_state_dict(ckpt)
    else:
        model.load_from_checkpoint(ckpt)
    model.eval()
    _state["model"] = model
    return model


def _pre(model, segment, device):
    """`pre` of the handlers (eval_gsr_voicefixer.py:19-25): segment (n,) -> sp, mel, x (1, 1, n); a stack of equal-length
    segments (k, n) -> the same for a batch of k."""
    x = segment if isinstance(segment, torch.Tensor) else torch.tensor(segment)
    x = (x[None, None, ...] if x.dim() == 1 else x[:, None, :]).to(device)
    sp, _, _ = model.f_helper.wav_to_spectrogram_phase(x)
    mel_orig = model.mel(sp.permute(0, 1, 3, 2)).permute(0, 1, 3, 2)
    return sp, mel_orig, x


class _PinnedPool:
    """Page-locked staging buffers, reused across segments and files.  `torch.empty(..., pin_memory=True)` / `.pin_memory()` per
    segment hands the buffer back to torch's caching host allocator, which can only recycle it once the copy that used it has
    finished -- every other file of a back-to-back evaluation found its block still busy and paid a fresh hipHostMalloc (10-20 ms:
    round 6, the handler's calls on one file alternated between 40 and 59 ms).  Here a buffer is taken with `take`, given back with
    `give` together with the event behind its last use, and handed out again only when that event has completed."""

    def __init__(self):
        self.free = []          # (tensor, event or None)

    def take(self, n, dtype):
        for i, (t, ev) in enumerate(self.free):
            if t.dtype == dtype and t.numel() >= n and (ev is None or ev.query()):
                del self.free[i]
                return t
        return torch.empty((max(int(n), 1),), dtype=dtype, pin_memory=torch.cuda.is_available())

    def give(self, t, ev=None):
        self.free.append((t, ev))
        if len(self.free) > 16:       # bounded: drop the smallest
            self.free.sort(key=lambda p: -p[0].numel())
            self.free.pop()


_PINNED = _PinnedPool()


def _staged_upload(host, n, device):
    """host[:n] (a pool buffer) -> a tensor on `device`; the buffer goes back to the pool behind the copy."""
    if torch.device(device).type != "cuda":
        out = host[:n].clone()
        _PINNED.give(host)
        return out
    out = host[:n].to(device, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    _PINNED.give(host, ev)
    return out


class _WavReader:
    """librosa.load(path, 44100) semantics of `load_wav`, streamed: the PCM16 frames of a 44.1 kHz file are read one segment
    at a time, so the GPU starts on the first 60 s while the host is still reading the rest.  Anything else (other sample
    widths, other rates: polyphase resampling needs the whole signal) falls back to `load_wav`, then slices."""

    def __init__(self, path, sample_rate=44100):
        self.f = wave.open(path, "rb")
        self.ch, width, sr = self.f.getnchannels(), self.f.getsampwidth(), self.f.getframerate()
        self.n = self.f.getnframes()
        self.whole = None
        self.pos = 0
        if width != 2 or sr != sample_rate:
            self.f.close()
            self.f = None
            self.whole = load_wav(path, sample_rate)
            self.n = self.whole.shape[0]

    def __len__(self):
        return self.n

    def read(self, count):
        """next `count` samples (fewer at the end of the file) as mono float32, the values `load_wav` returns"""
        if self.whole is not None:
            x = self.whole[self.pos:self.pos + count]
        else:
            raw = self.f.readframes(count)
            x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
            x = x.reshape(-1, self.ch).mean(axis=1) if self.ch > 1 else x
        self.pos += x.shape[0]
        if x.shape[0] < count:
            # the data chunk ended: `load_wav` would have returned exactly the frames that are there, whatever the header
            # promised (a truncated file, a streamed header) -- the segment loop ends here instead of reading empty segments
            self.n = self.pos
        return x

    def read_device(self, count, device):
        """`read` for the handlers: the same float32 values as a tensor on `device`.  Mono PCM16 at the target rate (what the
        evaluation sets hold) travels as the file's 2-byte samples and is widened on the device -- int16 -> float32 -> / 32768 is exact
        in both places, so the values are bit-identical to `read`'s -- which leaves the host one copy into pinned memory instead
        of two float32 passes over twice the bytes (round 5: the host's part of a 150-s file was 3-25 ms depending on the box)."""
        if self.whole is None and self.ch == 1:
            raw = self.f.readframes(count)
            n = len(raw) // 2
            self.pos += n
            if n < count:
                self.n = self.pos
            if n == 0:
                return torch.empty((0,), dtype=torch.float32, device=device)
            host = _PINNED.take(n, torch.int16)
            host[:n].numpy()[:] = np.frombuffer(raw, dtype="<i2")      # one copy, into page-locked memory
            return _staged_upload(host, n, device).to(torch.float32) / 32768.0
        x = self.read(count)
        host = _PINNED.take(x.shape[0], torch.float32)
        host[:x.shape[0]].numpy()[:] = x
        return _staged_upload(host, x.shape[0], device)

    def close(self):
        if self.f is not None:
            self.f.close()


class _WavWriter:
    """`save_wave` for a file that is produced segment by segment on the device: the PCM16 conversion of tools/file/wav.py
    ((x.astype(float64) * 2**15).astype(np.short), incl. its wrap of exactly +1.0) runs on the device, the 2-byte samples
    travel into pinned memory behind the segment's kernels, and a segment is written to the file while the next one is
    being computed."""

    def __init__(self, fname, sample_rate=44100):
        # like the reference's save_wave at the END of handler(), the output file only changes when the whole file went
        # through: the segments go to `fname + ".part"`, renamed by close(ok=True), removed otherwise
        self.fname, self.part = fname, fname + ".part"
        self.f = wave.open(self.part, "wb")
        self.f.setnchannels(1)
        self.f.setsampwidth(2)
        self.f.setframerate(sample_rate)
        self.pending = []

    def put(self, seg):
        """seg: (n,) float32 on the device; enqueues conversion + download, writes what has arrived before."""
        pcm = (seg.double() * 2 ** 15).to(torch.int32).to(torch.int16)   # float64 -> int32 truncation, then the 16-bit wrap of numpy
        buf = _PINNED.take(pcm.numel(), torch.int16)
        host = buf[:pcm.numel()]
        host.copy_(pcm, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending.append((host, ev, buf))
        self.flush(block=False)

    def flush(self, block=True):
        while self.pending and (block or self.pending[0][1].query()):
            host, ev, buf = self.pending.pop(0)
            ev.synchronize()
            self.f.writeframes(host.numpy())     # (the buffer itself: writeframes takes any bytes-like object, no copy)
            _PINNED.give(buf)

    def close(self, ok=True):
        import os
        try:
            if ok:
                self.flush(block=True)
        finally:
            self.f.close()
            if ok:
                os.replace(self.part, self.fname)
            elif os.path.exists(self.part):
                os.remove(self.part)


def _file_flags_ok(model, writer):
    """The checks of a whole file, once its last segment is enqueued: wait for the device (the writer's last download), then
    to_log's assert (pytorch_util.py:158; raises -- the output file is not touched, as in the reference where the assert fires
    before save_wave) and the 16-bit mode's saturation flag from the sticky device flags (-> False: the caller repeats the
    file on the split-bf16 twin)."""
    from . import _lib
    writer.flush(block=True)
    flags = model.engine.take_flags()
    if flags & _lib.FLAG_NEGATIVE_INPUT:
        raise AssertionError("to_log: input has negative values")
    return not (flags & _lib.FLAG_F16_SATURATED)


def _warn_peaks(peaks, input):
    """eval_gsr_voicefixer.py:68-70's warning, once per file from the per-segment peaks."""
    if peaks and bool((torch.stack(peaks) > 1.0).any()):
        print("Warning: Exceed energy limit,", input)


def _peak_normalise(out):
    """`if max|out| > 1: out = out / max|out|` (eval_gsr_voicefixer.py:68-70) without the host round trip of the compare:
    the same values bit for bit (x / p where p > 1, x otherwise); the peak is kept for the warning.  out (k, 1, n): every
    segment of a batch has its own peak (the reference handles one segment per call)."""
    peak = torch.amax(torch.abs(out), dim=(1, 2), keepdim=True)
    return torch.where(peak > 1.0, out / peak, out), peak.max()


def _is_out_of_memory(e):
    m = str(e).lower()
    return "out of memory" in m or "hipmalloc" in m or "hiperroroutofmemory" in m


def handler_gsr_voicefixer(input, output, target, ckpt, device, needrefresh=False, meta={}):
    """eval_gsr_voicefixer.py:37-77.  Same calls on the model surface, in the same order, as the reference; what differs is
    WHEN the host waits: the reference's two host syncs per segment (to_log's assert inside the model call, the peak
    compare) are checked once per file from device-side flags, the input file is read and the output file written one
    segment at a time beside the GPU work.  The metric values of a target (four device-to-host copies per segment, as
    in the reference) are the remaining per-segment waits."""
    if needrefresh or _state["model"] is None:
        refresh_model(ckpt, models.VoiceFixer)
    model = _state["model"].to(device)
    metrics = {}
    unify = meta.get("unify_energy", False)

    def run(model, output):
        nonlocal metrics
        # every file starts with clean flags: an exception in the middle of the previous file (frame mismatch, a short tail
        # segment, out of memory) skipped its flag check and would otherwise hand this file a stale saturation / negative bit
        model.engine.take_flags()
        reader = _WavReader(input, 44100)
        tgt = load_wav(target, sample_rate=44100) if target is not None else None
        writer = _WavWriter(output, 44100)
        peaks = []
        seg_length = 44100 * SEG_SECONDS
        break_point = seg_length
        done = False
        try:
            while break_point < len(reader) + seg_length:
                # Without a target there are no per-segment metrics, and a segment's result does not depend on the batch it is
                # in (the kernels' sums are per clip): the FULL segments of the file go through the same calls up to
                # MAX_SEGMENT_BATCH at a time -- the same bytes in the file, fewer and larger launches (round 5).  With a
                # target the loop stays the reference's: one segment per iteration, its metrics replace the previous ones.
                k = 1
                if tgt is None:
                    k = max(1, min(MAX_SEGMENT_BATCH, (len(reader) - (break_point - seg_length)) // seg_length))
                segment = reader.read_device(k * seg_length, device)
                if segment.shape[0] == 0:      # the header promised more frames than the file holds
                    break
                full = segment.shape[0] // seg_length if k > 1 else 0
                pieces = ([segment[:full * seg_length].reshape(full, seg_length)] if full else []) + \
                         ([segment[full * seg_length:]] if segment.shape[0] > full * seg_length else [])
                queue = list(pieces)
                while queue:
                    piece = queue.pop(0)
                    try:
                        _, mel_noisy, seg_t = _pre(model, piece.contiguous(), device)
                        out_model = model(mel_noisy, check=False)
                    except RuntimeError as e:
                        # a batch of full segments needs k times the workspace of one (about 0.8 GB per 60-s segment): on a
                        # smaller or shared GPU fall back to the reference's one segment per call instead of failing the file
                        if piece.dim() == 2 and piece.shape[0] > 1 and _is_out_of_memory(e):
                            torch.cuda.empty_cache()
                            queue = [piece[j:j + 1] for j in range(piece.shape[0])] + queue
                            continue
                        raise
                    denoised_mel = from_log(out_model["mel"])
                    if unify:
                        denoised_mel, mel_noisy = amp_to_original_f(mel_sp_est=denoised_mel, mel_sp_target=mel_noisy)
                    if tgt is not None:
                        # like the reference (eval_gsr_voicefixer.py:56-64) the estimate and the target segment must have the
                        # same number of frames: a shorter / longer target file raises instead of being trimmed silently
                        _, target_mel, _ = _pre(model, tgt[break_point - seg_length:break_point], device)
                        if target_mel.shape != denoised_mel.shape:
                            raise RuntimeError("The size of tensor a (%d) must match the size of tensor b (%d) at non-singleton "
                                               "dimension 2" % (denoised_mel.shape[2], target_mel.shape[2]))
                        m_lsd, m_lin = device_metrics(model.engine, denoised_mel.contiguous(), target_mel.contiguous())
                        _, m_log = device_metrics(model.engine, out_model["mel"].contiguous(), to_log(target_mel))
                        # non-log SiSpec is defined on from_log(model output) (eval_gsr_voicefixer.py:62), i.e. before unify_energy
                        if unify:
                            _, m_lin = device_metrics(model.engine, from_log(out_model["mel"]).contiguous(), target_mel.contiguous())
                        metrics = {"mel-lsd": m_lsd, "mel-sispec": m_log, "mel-non-log-sispec": m_lin,
                                   "mel-ssim": float(ssim(denoised_mel, target_mel))}
                    try:
                        out = model.vocoder(denoised_mel, check=False)
                    except RuntimeError as e:      # (the vocoder's workspace is the larger one: the same fallback)
                        if piece.dim() == 2 and piece.shape[0] > 1 and _is_out_of_memory(e):
                            del out_model, denoised_mel, mel_noisy
                            torch.cuda.empty_cache()
                            queue = [piece[j:j + 1] for j in range(piece.shape[0])] + queue
                            continue
                        raise
                    out, peak = _peak_normalise(out)
                    peaks.append(peak)
                    out, _ = trim_center(out, seg_t)
                    for j in range(out.shape[0]):
                        writer.put(out[j, 0])
                    break_point += seg_length * out.shape[0]
            done = _file_flags_ok(model, writer)
        finally:
            reader.close()
            writer.close(ok=done)
        return peaks if done else None

    with torch.no_grad():
        peaks = run(model, output)
        if peaks is None:
            # the 16-bit vocoder clamped an activation somewhere in this file: the whole file again on split-bf16 operands
            import warnings
            warnings.warn("16-bit vocoder: an activation left the fp16 range; %s is restored again with split-bf16 operands" % input)
            strict = models.VoiceFixer(model.hp, channels=model.channels, type_target=model.type_target,
                                       engine=model.engine.strict_twin())
            peaks = run(strict, output)
            if peaks is None:
                raise RuntimeError("the split-bf16 twin reported a clamped activation")
        _warn_peaks(peaks, input)
    return metrics


def handler_ssr_unet(input, output, target, ckpt, device, needrefresh=False, meta={}):
    """eval_ssr_unet.py:95-143 (also serves eval_gsr_unet.py); host waits as in handler_gsr_voicefixer."""
    if needrefresh or _state["model"] is None:
        refresh_model(ckpt, models.SSR_UNet)
    model = _state["model"].to(device)
    metrics = {}
    with torch.no_grad():
        # every file starts with clean flags: an exception in the middle of the previous file (frame mismatch, a short tail
        # segment, out of memory) skipped its flag check and would otherwise hand this file a stale saturation / negative bit
        model.engine.take_flags()
        reader = _WavReader(input, 44100)
        tgt = load_wav(target, sample_rate=44100) if target is not None else None
        writer = _WavWriter(output, 44100)
        peaks = []
        seg_length = 44100 * SEG_SECONDS
        break_point = seg_length
        done = False
        try:
            while break_point < len(reader) + seg_length:
                segment = reader.read_device(seg_length, device)
                if segment.shape[0] == 0:      # the header promised more frames than the file holds
                    break
                sp, _, seg_t = _pre(model, segment, device)
                out = model(sp, seg_t)["wav"]
                if tgt is not None:
                    sp_o, _, _ = model.f_helper.wav_to_spectrogram_phase(out)
                    mel_out = model.mel(sp_o.permute(0, 1, 3, 2)).permute(0, 1, 3, 2)
                    _, target_mel, _ = _pre(model, tgt[break_point - seg_length:break_point], device)
                    if target_mel.shape != mel_out.shape:
                        raise RuntimeError("The size of tensor a (%d) must match the size of tensor b (%d) at non-singleton "
                                           "dimension 2" % (mel_out.shape[2], target_mel.shape[2]))
                    m_lsd, m_lin = device_metrics(model.engine, mel_out.contiguous(), target_mel.contiguous())
                    _, m_log = device_metrics(model.engine, to_log(mel_out), to_log(target_mel))
                    metrics = {"mel-lsd": m_lsd, "mel-sispec": m_log, "mel-non-log-sispec": m_lin,
                               "mel-ssim": float(ssim(mel_out, target_mel))}
                out, peak = _peak_normalise(out)
                peaks.append(peak)
                out, _ = trim_center(out, seg_t)
                writer.put(out[0, 0])
                break_point += seg_length
            done = _file_flags_ok(model, writer)      # (no fp16 arithmetic on this path: only to_log's assert can fire)
        finally:
            reader.close()
            writer.close(ok=done)
        _warn_peaks(peaks, input)
    return metrics


# names the eval scripts use
handler = handler_gsr_voicefixer
