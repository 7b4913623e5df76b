"""Long-audio chunkers: the reference's two `LambdaOverlapAdd` classes, batched for one GPU.

* `LambdaOverlapAdd`        -- tools/dsp/overlapadd.py:337-480: windowed overlap-add.  The signal is cut into
  `window_size` chunks every `hop_size` samples (with `window_size` zeros in front and behind, so the first
  chunks are partly or wholly silence), the network runs on every chunk, every output is multiplied by the
  synthesis window and the chunks are summed back in place.
* `LambdaOverlapAddBoxcar`  -- tools/dsp/overlapadd_boxcar.py:338-513 (same class name there): non-overlapping
  frames of `window_size` that see `in_margin` samples of context on each side; the context is cut from the
  network output and the frames are concatenated.  The first and the last frame have one-sided context, and a
  ragged last frame is run at its own length.

Same constructor arguments, `forward(x, key)` / `ola_forward(x, key)` and results as the reference classes.
What differs is the execution: the reference calls `nnet` once per chunk in a Python loop ("for loop to spare
memory", overlapadd.py:436); chunks are independent, so here every group of equal-length chunks is ONE batched
call of `nnet` (a 288-GB GPU holds minutes of 1-s chunks at once; `max_batch` bounds a call), and the cutting and
stitching are two HBM-bound kernels of libvfx (`vfx_chunk_gather`, `vfx_chunk_ola`) instead of
`F.unfold`/`F.fold`.  The models of this package are deterministic and clip-independent, so batching does not
change a chunk's result (tests/test_gpu_surface.py checks bit-equality against the one-chunk-at-a-time order).

`nnet(x: (N, channels, n)) -> {key: (N, n_src, n)}`.  Source re-ordering between chunks
(`_reorder_sources`, overlapadd.py:499-531) is the identity for one source, the only case the restoration
models have; `n_src > 1` is refused.  Window names: the reference's default "hanning" is scipy's periodic Hann
(the spelling newer scipy releases dropped); `window=None` -- which the reference constructors cannot take
(overlapadd.py:411 calls `.type_as` on None) -- selects the un-windowed branch `frame / (window_size / hop_size)`.
"""
import math

import numpy as np
import torch


def _window(name, n):
    if name in ("hann", "hanning", "han"):
        return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n, dtype=np.float64) / n)).astype(np.float32)
    if name in ("boxcar", "box", "ones", "rect", "rectangular"):
        return np.ones(n, np.float32)
    from scipy.signal import get_window
    return get_window(name, n).astype(np.float32)


class _ChunkerBase:
    def __init__(self, nnet, n_src, window_size, window, reorder_chunks, enable_grad, device, engine, max_batch):
        assert window_size % 2 == 0, "Window size must be even"
        if n_src is not None and n_src != 1:
            raise NotImplementedError("chunk re-ordering across sources (n_src > 1) is outside the restoration path")
        self.nnet = nnet
        self.n_src = n_src
        self.window_size = window_size
        self.in_channels = getattr(nnet, "in_channels", None)
        self.engine = engine if engine is not None else getattr(nnet, "engine", None)
        if self.engine is None:
            raise ValueError("no libvfx engine: pass engine= or an nnet that carries one")
        self.device = self.engine.device
        self.use_window = bool(window)
        self.window = torch.from_numpy(_window(window, window_size)).to(self.device) if window else None
        self.reorder_chunks = reorder_chunks    # identity for one source
        self.enable_grad = enable_grad          # inference only: never records a graph
        self.max_batch = int(max_batch)

    def _run(self, chunks, key):
        """nnet on (N, channels, n) in calls of at most max_batch chunks -> (N, n_src, n)."""
        outs = []
        for i in range(0, chunks.shape[0], self.max_batch):
            frame = self.nnet(chunks[i:i + self.max_batch])[key]
            assert frame.ndim == 3, "nnet should return (batch, n_src, time)"
            if self.n_src is not None:
                assert frame.shape[1] == self.n_src, "nnet should return (batch, n_src, time)"
            if frame.shape[1] != 1:
                raise NotImplementedError("nnet returned %d sources; one is supported" % frame.shape[1])
            outs.append(frame)
        return outs[0] if len(outs) == 1 else torch.cat(outs, 0)

    def forward(self, x, key="wav"):
        with torch.no_grad():
            return self.ola_forward(x, key=key)

    __call__ = forward

    @property
    def sample_rate(self):
        return self.nnet.sample_rate

    def _separate(self, wav, *args, **kwargs):
        return self.forward(wav, *args, **kwargs)


class LambdaOverlapAdd(_ChunkerBase):
    """tools/dsp/overlapadd.py:337-480."""

    def __init__(self, nnet, n_src, window_size, hop_size=None, window="hanning", reorder_chunks=True,
                 enable_grad=False, device=None, engine=None, max_batch=64):
        super().__init__(nnet, n_src, window_size, window, reorder_chunks, enable_grad, device, engine, max_batch)
        self.hop_size = hop_size if hop_size is not None else window_size // 2

    def ola_forward(self, x, key="wav"):
        assert x.ndim == 3
        batch, channels, n_frames = x.shape
        W, hop = self.window_size, self.hop_size
        n_chunks = (n_frames + W) // hop + 1            # F.unfold(kernel W, padding W, stride hop), :421-428
        flat = x.reshape(batch * channels, n_frames)
        chunks = self.engine.chunk_gather(flat, W, hop, W, n_chunks)             # (batch*ch, n_chunks, W)
        chunks = chunks.view(batch, channels, n_chunks, W).permute(0, 2, 1, 3).reshape(batch * n_chunks, channels, W)
        frames = self._run(chunks, key)                                           # (batch*n_chunks, 1, W)
        frames = frames.reshape(batch, n_chunks, W)
        out = self.engine.chunk_ola(frames, self.window, 1.0 / (W / hop), hop, W, n_frames)   # :455-471
        return out.reshape(batch, 1, n_frames)


class LambdaOverlapAddBoxcar(_ChunkerBase):
    """tools/dsp/overlapadd_boxcar.py:338-513 (`LambdaOverlapAdd` of that module)."""

    def __init__(self, nnet, n_src, window_size, in_margin, window="hanning", reorder_chunks=True,
                 enable_grad=False, device=None, engine=None, max_batch=64):
        super().__init__(nnet, n_src, window_size, window, reorder_chunks, enable_grad, device, engine, max_batch)
        self.hop_size = window_size
        self.in_margin = in_margin

    def ola_forward(self, x, key="wav"):
        assert x.ndim == 3
        batch, channels, n_frames = x.shape
        W, M = self.window_size, self.in_margin
        last = n_frames - (n_frames // W) * W
        n_chunks = int(math.ceil(n_frames / W))
        flat = x.reshape(batch * channels, n_frames)
        # frame k with its margins: x[k*W - M : (k+1)*W + M], zeros outside the signal (:436-452)
        full = self.engine.chunk_gather(flat, W + 2 * M, W, M, n_chunks)
        full = full.view(batch, channels, n_chunks, W + 2 * M)
        frames = torch.zeros((batch, n_chunks, W), device=self.device, dtype=torch.float32)
        # first frame: no context in front (:458-461)
        frames[:, 0] = self._run(full[:, :, 0, M:].contiguous(), key)[:, 0, :W]
        if n_chunks > 1:
            if last != 0:   # ragged last frame runs at its own length (:462-466)
                f = self._run(full[:, :, n_chunks - 1, :M + last].contiguous(), key)
                frames[:, n_chunks - 1, :last] = f[:, 0, M:]
            else:           # :467-470
                f = self._run(full[:, :, n_chunks - 1, :M + W].contiguous(), key)
                frames[:, n_chunks - 1] = f[:, 0, M:]
        if n_chunks > 2:    # all inner frames in one batch (:471-476)
            mid = full[:, :, 1:n_chunks - 1].permute(0, 2, 1, 3).reshape(batch * (n_chunks - 2), channels, W + 2 * M)
            f = self._run(mid.contiguous(), key)
            frames[:, 1:n_chunks - 1] = f[:, 0, M:M + W].reshape(batch, n_chunks - 2, W)
        out = self.engine.chunk_ola(frames, self.window, 1.0, W, 0, n_frames)    # :494-507
        return out.reshape(batch, 1, n_frames)
