"""CPU: host-side logic and the C-ABI library surface (no compute calls without a GPU)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from voicefixer_main_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "vfx.h")).read()
    declared = set(re.findall(r"\b(vfx_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found in include/vfx.h"
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), "libvfx.so does not export %s" % name
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    # ... and NOTHING else with C linkage: the product library is the reference-surface entry points + the profile hooks; the
    # kernel-level test entry points live in libvfx_test.so (include/vfx_test.h), which the product never loads
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    c_syms = {l.split()[-1] for l in nm.splitlines() if len(l.split()) == 3 and l.split()[1] in "TW" and l.split()[-1].startswith("vfx_")}
    assert c_syms == declared, sorted(c_syms ^ declared)
    thdr = open(os.path.join(ROOT, "include", "vfx_test.h")).read()
    tdecl = set(re.findall(r"\b(vfx_[a-z0-9_]+)\s*\(", thdr))
    assert tdecl and not (tdecl & declared) and tdecl == set(_lib.TEST_SIGNATURES), (tdecl ^ set(_lib.TEST_SIGNATURES))
    tlib = _lib.load_test()
    for name in sorted(tdecl):
        assert hasattr(tlib, name), "libvfx_test.so does not export %s" % name
    for f in ("models.py", "handlers.py", "dist.py", "chunker.py", "simulate.py", "synth.py"):
        assert "load_test" not in open(os.path.join(ROOT, "voicefixer_main_amd", f)).read(), f


def test_default_config_and_error_channel():
    from voicefixer_main_amd import _lib
    lib = _lib.load()
    cfg = _lib.VfxConfig()
    assert lib.vfx_default_config(cfg) == 0
    assert (cfg.sample_rate, cfg.n_fft, cfg.hop, cfg.n_mels) == (44100, 2048, 441, 128)
    assert list(cfg.voc_scales)[:4] == [7, 7, 3, 3] and int(np.prod(list(cfg.voc_scales)[:4])) == cfg.hop
    assert lib.vfx_load_tensor(None, 0, b"x", None, None, 0) != 0            # NULL handle -> error, not a crash
    assert b"NULL" in lib.vfx_last_error()


def test_product_path_refuses_cpu():
    from voicefixer_main_amd.engine import Engine
    with pytest.raises(RuntimeError):
        Engine("cpu")


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "voicefixer_main_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f


def test_wav_io_roundtrip_and_resample(tmp_path):
    from voicefixer_main_amd import handlers
    x = (np.sin(np.linspace(0, 200, 44100)) * 0.5).astype(np.float32)
    p = str(tmp_path / "a.wav")
    handlers.save_wave(x, p)
    y = handlers.load_wav(p)
    assert y.shape == x.shape and np.abs(y - x).max() < 2.0 / 32768
    import wave
    with wave.open(str(tmp_path / "b.wav"), "wb") as f:            # 22.05 kHz stereo -> 44.1 kHz mono
        f.setnchannels(2); f.setsampwidth(2); f.setframerate(22050)
        f.writeframes((np.stack([x[::2], x[::2]], 1) * 32767).astype("<i2").tobytes())
    z = handlers.load_wav(str(tmp_path / "b.wav"))
    assert abs(len(z) - 44100) <= 2


def test_streaming_wav_reader_equals_load_wav(tmp_path):
    """handlers._WavReader (segment-by-segment reads) returns, concatenated, exactly what load_wav returns -- mono / stereo PCM16
    at 44.1 kHz through its streaming path, 8- / 32-bit and 48 kHz files through its whole-file fallback -- and ends at the last
    frame the file really holds when the header promises more (a truncated file)."""
    import wave
    from voicefixer_main_amd import handlers
    rng = np.random.default_rng(3)

    def write(path, ch, width, sr, n):
        x = rng.uniform(-0.9, 0.9, (n, ch))
        with wave.open(path, "wb") as f:
            f.setnchannels(ch); f.setsampwidth(width); f.setframerate(sr)
            if width == 2:
                f.writeframes((x * 32767).astype("<i2").tobytes())
            elif width == 4:
                f.writeframes((x * 2147483647).astype("<i4").tobytes())
            else:
                f.writeframes(((x * 127) + 128).astype(np.uint8).tobytes())

    for i, (ch, width, sr, n) in enumerate([(1, 2, 44100, 10000), (2, 2, 44100, 10001), (1, 1, 44100, 5000), (2, 4, 44100, 5000),
                                             (1, 2, 48000, 9600), (2, 2, 22050, 4410)]):
        p = str(tmp_path / ("r%d.wav" % i))
        write(p, ch, width, sr, n)
        want = handlers.load_wav(p, 44100)
        r = handlers._WavReader(p, 44100)
        assert len(r) == want.shape[0]
        parts = []
        while sum(a.shape[0] for a in parts) < len(r):
            parts.append(r.read(3000))
            assert parts[-1].shape[0] > 0
        r.close()
        got = np.concatenate(parts)
        assert got.dtype == np.float32 and np.array_equal(got, want), (ch, width, sr)
    # truncated data chunk: the header says 10000 frames, the file holds 6500
    p = str(tmp_path / "t.wav")
    write(p, 1, 2, 44100, 10000)
    raw = open(p, "rb").read()
    open(p, "wb").write(raw[:44 + 2 * 6500])
    r = handlers._WavReader(p, 44100)
    assert len(r) == 10000
    parts = []
    while sum(a.shape[0] for a in parts) < len(r):
        seg = r.read(3000)
        if seg.shape[0] == 0:
            break
        parts.append(seg)
    r.close()
    assert sum(a.shape[0] for a in parts) == 6500 and len(r) == 6500 and [a.shape[0] for a in parts] == [3000, 3000, 500]


def test_handler_glue_matches_oracle_glue():
    from oracle import pipeline
    from voicefixer_main_amd import handlers
    a, b = torch.arange(110.0)[None, None], torch.zeros(1, 1, 100)
    t, _ = handlers.trim_center(a, b)
    t2, _ = pipeline.trim_center(a.numpy(), b.numpy())
    assert np.array_equal(t.numpy(), t2)
    est = torch.rand(2, 1, 6, 128) + 0.1
    tgt = torch.rand(2, 1, 6, 128) + 0.1
    s, _ = handlers.amp_to_original_f(est, tgt)
    s2, _ = pipeline.amp_to_original_f(est.numpy(), tgt.numpy())
    assert np.allclose(s.numpy(), s2, rtol=1e-5)
    assert float(handlers.lsd(est[:1], est[:1])) < 1e-6
    assert float(handlers.sispec(est, est)) > 60


def _stub_handler_run(monkeypatch, tmp_path, n_samples, truncate_to=None, oom_batches=0, seg=44100):
    """handler_gsr_voicefixer's segment / break_point bookkeeping on the CPU: a stub model whose `vocoder` returns the segment it was
    given with 2 x 882 samples of padding (trim_center crops them), a file writer without pinned memory.  The output file must
    hold exactly the samples the reader delivered."""
    import wave
    from voicefixer_main_amd import handlers
    rng = np.random.default_rng(n_samples)
    pcm = rng.integers(-20000, 20000, size=n_samples, dtype=np.int16)
    src, dst = str(tmp_path / "in.wav"), str(tmp_path / "out.wav")
    with wave.open(src, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(44100)
        f.writeframes(pcm.tobytes())
    if truncate_to is not None:                      # the header keeps promising n_samples frames
        raw = open(src, "rb").read()
        open(src, "wb").write(raw[:44 + 2 * truncate_to])
        pcm = pcm[:truncate_to]
    calls = {"batches": [], "oom_left": oom_batches}

    class Writer:
        def __init__(self, fname, sample_rate=44100):
            self.fname, self.parts = fname, []

        def put(self, seg_):
            self.parts.append((seg_.double() * 2 ** 15).to(torch.int32).to(torch.int16).numpy())

        def flush(self, block=True):
            pass

        def close(self, ok=True):
            if ok:
                handlers.save_wave(np.concatenate(self.parts).astype(np.float64) / 2 ** 15, self.fname)

    class FHelper:
        def wav_to_spectrogram_phase(self, x):
            model.last_x = x
            T = x.shape[-1] // 441 + 1
            return torch.ones((x.shape[0], 1, T, 1025)), None, None

    class Model:
        f_helper = FHelper()
        engine = type("E", (), {"take_flags": staticmethod(lambda: 0)})()

        def to(self, device):
            return self

        def mel(self, sp):                     # (B, 1, 1025, T) -> (B, 1, 128, T)
            return sp[:, :, :128, :]

        def __call__(self, mel, check=True):
            calls["batches"].append(mel.shape[0])
            if mel.shape[0] > 1 and calls["oom_left"] > 0:
                calls["oom_left"] -= 1
                raise RuntimeError("HIP out of memory. Tried to allocate 3.00 GiB")
            return {"mel": handlers.to_log(mel)}

        def vocoder(self, mel, check=True):
            return torch.nn.functional.pad(self.last_x, (882, 882))

    model = Model()
    monkeypatch.setattr(handlers, "_WavWriter", Writer)
    monkeypatch.setattr(handlers._WavReader, "read_device", lambda self, count, device: torch.from_numpy(np.ascontiguousarray(self.read(count))))
    monkeypatch.setattr(handlers, "SEG_SECONDS", seg // 44100)     # 1-s segments instead of 60-s ones
    monkeypatch.setitem(handlers._state, "model", model)
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda: None)
    handlers.handler_gsr_voicefixer(src, dst, None, "unused.ckpt", torch.device("cpu"))
    got = np.frombuffer(wave.open(dst, "rb").readframes(10 ** 9), dtype="<i2")
    return pcm, got, calls


def test_handler_segment_batches_bookkeeping(monkeypatch, tmp_path):
    """Without a target the handler runs the FULL segments of a file up to MAX_SEGMENT_BATCH at a time (handlers.py): 6.3 segments
    are calls of 4, 2 and the 0.3-segment tail; every sample appears once, in order."""
    pcm, got, calls = _stub_handler_run(monkeypatch, tmp_path, int(44100 * 6.3))
    assert calls["batches"] == [4, 2, 1]
    assert np.array_equal(pcm, got)
    # exactly k full segments: no tail call
    pcm, got, calls = _stub_handler_run(monkeypatch, tmp_path, 44100 * 5)
    assert calls["batches"] == [4, 1] and np.array_equal(pcm, got)


def test_handler_header_promises_more_frames_than_the_file_holds(monkeypatch, tmp_path):
    """A truncated file (the header says 9.5 segments, the data chunk ends after 5.25): the batched reads come back short, the loop
    ends at the real end of the data and the output holds exactly what was there."""
    pcm, got, calls = _stub_handler_run(monkeypatch, tmp_path, int(44100 * 9.5), truncate_to=int(44100 * 5.25))
    assert np.array_equal(pcm, got)
    assert sum(calls["batches"][:2]) == 5 and calls["batches"][-1] == 1


def test_handler_batched_segments_fall_back_on_out_of_memory(monkeypatch, tmp_path):
    """A batch of segments that does not fit (RuntimeError: out of memory) is retried one segment per call -- the reference's own
    loop -- instead of failing the file."""
    pcm, got, calls = _stub_handler_run(monkeypatch, tmp_path, int(44100 * 6.3), oom_batches=1)
    assert calls["batches"] == [4, 1, 1, 1, 1, 2, 1]
    assert np.array_equal(pcm, got)


def test_ssim_matches_the_skimage_algorithm():
    """handlers.ssim restates skimage.metrics.structural_similarity(win_size=7) (evaluation_proc/metrics.py:97-106;
    skimage is not installed): checked against the published algorithm written with scipy's uniform_filter."""
    from scipy.ndimage import uniform_filter
    from voicefixer_main_amd import handlers
    rng = np.random.default_rng(0)
    a = rng.random((2, 1, 40, 128)).astype(np.float32)
    b = (a + 0.1 * rng.normal(size=a.shape)).astype(np.float32)

    def sk(im1, im2, win=7, R=2.0):
        im1, im2 = im1.astype(np.float64), im2.astype(np.float64)
        NP = win * win
        cov_norm = NP / (NP - 1)
        ux, uy = uniform_filter(im1, size=win), uniform_filter(im2, size=win)
        uxx, uyy, uxy = uniform_filter(im1 * im1, size=win), uniform_filter(im2 * im2, size=win), uniform_filter(im1 * im2, size=win)
        vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
        C1, C2 = (0.01 * R) ** 2, (0.03 * R) ** 2
        S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
        pad = (win - 1) // 2
        return S[pad:-pad, pad:-pad].mean()
    got = handlers.ssim(torch.from_numpy(a), torch.from_numpy(b))
    assert got.shape == (2, 1, 1, 1)
    for i in range(2):
        assert abs(float(got[i, 0, 0, 0]) - sk(a[i, 0], b[i, 0])) < 1e-9
    assert abs(float(handlers.ssim(torch.from_numpy(a[:1]), torch.from_numpy(a[:1]))) - 1.0) < 1e-12


def test_weight_norm_folding():
    from voicefixer_main_amd.models import fold_weight_norm
    v = torch.randn(4, 3, 5)
    g = torch.rand(4, 1, 1) + 0.5
    ref = torch._weight_norm(v, g, 0)
    out = fold_weight_norm({"c.weight_g": g, "c.weight_v": v, "c.bias": torch.zeros(4)})
    assert set(out) == {"c.weight", "c.bias"} and torch.allclose(out["c.weight"], ref, atol=1e-6)


def test_sharded_step_without_process_group():
    """A single process (bench.py --workload sharded1024 at N = 1) is a world of one: no torch.distributed needed."""
    from voicefixer_main_amd import dist as vdist
    full = torch.arange(40.0).reshape(5, 8)
    back, t = vdist.sharded_step(lambda x: x + 1.0, full, 5, 8, torch.device("cpu"))
    assert torch.equal(back, full + 1.0) and vdist.live_ranks(torch.device("cpu")) == 1


def test_bench_refuses_a_world_size_that_disagrees_with_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, env=env,
                       timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_broadcast_state_dict_world_of_one():
    from voicefixer_main_amd import dist as vdist
    sd = {"w": torch.ones(2, 2)}
    assert vdist.broadcast_state_dict(sd, torch.device("cpu")) is sd


def test_shard_bounds():
    from voicefixer_main_amd.dist import shard_bounds
    assert shard_bounds(1024, 8) == [(i * 128, (i + 1) * 128) for i in range(8)]
    b = shard_bounds(10, 4)
    assert b == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert shard_bounds(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from voicefixer_main_amd import dist as vdist
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cpu")
n, L = 5, 64
full = torch.arange(n * L, dtype=torch.float32).reshape(n, L) if rank == 0 else None
mine = vdist.scatter_clips(full, n, L, dev)
lo, hi = vdist.shard_bounds(n, world)[rank]
assert mine.shape == (hi - lo, L) and float(mine[0, 0]) == lo * L
back = vdist.gather_clips(mine + 1.0, n, L, dev)
if rank == 0:
    assert torch.equal(back, full + 1.0)
assert vdist.selfcheck(dev)
# the timed step of bench.py --workload sharded1024 (BASELINE.json configs[3]): scatter -> restore -> gather
assert vdist.world_rank() == (world, rank) and vdist.live_ranks(dev) == world
back, t = vdist.sharded_step(lambda x: x * 3.0 + 1.0, full, n, L, dev)
if rank == 0:
    assert torch.equal(back, full * 3.0 + 1.0)
else:
    assert back is None
assert set(t) == {"scatter_ms", "restore_ms", "gather_ms"} and all(v >= 0 for v in t.values())
# unequal shards with an EMPTY rank (fewer clips than ranks): rank 1 gets nothing, restores nothing, sends nothing
full1 = torch.arange(L, dtype=torch.float32).reshape(1, L) if rank == 0 else None
mine = vdist.scatter_clips(full1, 1, L, dev)
assert mine.shape == ((1, L) if rank == 0 else (0, L))
back, t = vdist.sharded_step(lambda x: x - 2.0, full1, 1, L, dev)
if rank == 0:
    assert torch.equal(back, full1 - 2.0)
# three clips on two ranks: shards of 2 and 1
full3 = torch.arange(3 * L, dtype=torch.float32).reshape(3, L) if rank == 0 else None
back = vdist.restore_sharded(lambda x: x * 0.5, full3, 3, L, dev)
if rank == 0:
    assert torch.equal(back, full3 * 0.5)
# weights: rank 0 holds the state_dict, one flat broadcast, every rank ends up with the same tensors (SURVEY 8e)
sd = None
if rank == 0:
    g = torch.Generator().manual_seed(3)
    sd = {"a.weight": torch.randn(4, 3, 3, 3, generator=g), "a.bias": torch.randn(4, generator=g),
          "bn.num_batches_tracked": torch.tensor(7), "s": torch.randn((), generator=g).double()}
got = vdist.broadcast_state_dict(sd, dev)
g = torch.Generator().manual_seed(3)
want = {"a.weight": torch.randn(4, 3, 3, 3, generator=g), "a.bias": torch.randn(4, generator=g), "s": torch.randn((), generator=g)}
assert set(got) == set(want) and all(got[k].shape == want[k].shape and torch.equal(got[k], want[k].float()) for k in want), rank
# a test set of clips of UNEQUAL lengths (SURVEY 8e: sorted by length, dealt round-robin; equal lengths batch together):
# 5 clips of 3 lengths on 2 ranks; the stand-in engine marks every element with its clip's batch size, so the batching shows
calls = []
def eng(x):
    calls.append(tuple(x.shape))
    return x * 2.0 + float(x.shape[0])
lens = [100, 37, 100, 64, 37]
clips = [torch.arange(L, dtype=torch.float32) + 1000.0 * i for i, L in enumerate(lens)] if rank == 0 else None
assert vdist.deal_by_length(lens, 2) == [0, 1, 1, 0, 0]          # by length: 100, 100, 64, 37, 37 -> ranks 0, 1, 0, 1, 0
back = vdist.restore_sharded_lengths(eng, clips, dev)
assert sorted(calls) == ([(1, 37), (1, 64), (1, 100)] if rank == 0 else [(1, 37), (1, 100)]), (rank, calls)
if rank == 0:
    assert [int(b.shape[0]) for b in back] == lens
    assert all(torch.equal(b, c * 2.0 + 1.0) for b, c in zip(back, clips))
else:
    assert back is None
# equal lengths on one rank DO share a call: 4 clips of one length -> two per rank, one call each; max_batch splits a bucket
calls.clear()
clips = [torch.full((50,), float(i)) for i in range(4)] if rank == 0 else None
back = vdist.restore_sharded_lengths(eng, clips, dev)
assert calls == [(2, 50)], (rank, calls)
if rank == 0:
    assert all(torch.equal(b, c * 2.0 + 2.0) for b, c in zip(back, clips))
calls.clear()
back = vdist.restore_sharded_lengths(eng, clips, dev, max_batch=1)
assert calls == [(1, 50), (1, 50)]
# fewer clips than ranks: an EMPTY rank takes part in nothing but the length broadcast
calls.clear()
clips = [torch.arange(9, dtype=torch.float32)] if rank == 0 else None
back = vdist.restore_sharded_lengths(eng, clips, dev)
assert calls == ([(1, 9)] if rank == 0 else [])
if rank == 0:
    assert torch.equal(back[0], clips[0] * 2.0 + 1.0)
# round 5: an engine function that takes a length vector (checked_restore: fn(x, lengths), fn.bucket_key) gets ONE padded batch
# per bucket -- 6 clips of 6 lengths, bucket = length // 50: rank 0 holds 120, 101, 45 -> calls (2, 120) + lengths, (1, 45)
calls.clear()
def eng_v(x, lengths=None):
    calls.append((tuple(x.shape), None if lengths is None else tuple(lengths)))
    y = x * 2.0
    if lengths is not None:
        for j, n in enumerate(lengths):
            assert float(x[j, n:].abs().sum()) == 0.0            # the padding of a padded batch is zeros
    return y
eng_v.bucket_key = lambda n: n // 50
lens = [120, 110, 101, 47, 40, 45]
clips = [torch.arange(L, dtype=torch.float32) + 1.0 + 1000.0 * i for i, L in enumerate(lens)] if rank == 0 else None
back = vdist.restore_sharded_lengths(eng_v, clips, dev)
want_calls = [((2, 120), (120, 101)), ((1, 45), None)] if rank == 0 else [((1, 110), None), ((2, 47), (47, 40))]
assert calls == want_calls, (rank, calls)
if rank == 0:
    assert [int(b.shape[0]) for b in back] == lens and all(torch.equal(b, c * 2.0) for b, c in zip(back, clips))
# an exception on ONE rank (to_log's assert, a twin that cannot be created) is raised on EVERY rank instead of leaving the
# others waiting in the gather
def eng_bad(x, lengths=None):
    if rank == 1:
        raise AssertionError("to_log: input has negative values")
    return x
try:
    vdist.restore_sharded_lengths(eng_bad, clips, dev)
    raise SystemExit("rank %%d: no exception" %% rank)
except AssertionError as e:
    assert rank == 1 and "negative" in str(e)
except RuntimeError as e:
    assert rank == 0 and "another rank" in str(e)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_scatter_gather_two_ranks_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29617", str(script)],
                         capture_output=True, text=True, env=env, timeout=240)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("ok") == 2


@pytest.mark.parametrize("workload,launcher", [("gsr16x10", "torchrun"), ("sharded1024", "torchrun"), ("gsr16x10", "self-spawn")])
def test_bench_rank_plumbing_dry_run_two_ranks(workload, launcher):
    """bench.py's OWN N > 1 branch has never run on hardware (no multi-GPU box in rounds 1-4): `--dry-run-cpu` runs exactly
    that code -- process group, one-flat-broadcast of the weights, per-rank clips / scatter + gather, barrier + max-over-ranks
    clock, rank 0's JSON line -- over gloo with an identity stub where the libvfx handle stands, launched the way the driver
    launches it (`python -m torch.distributed.run ... bench.py --gpus N`) and the way `bench.py --gpus N` spawns itself."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MASTER_ADDR"] = "127.0.0.1"
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run-cpu", "--workload", workload,
            "--clips", "3", "--seconds", "0.2"]
    port = {"gsr16x10": 29641, "sharded1024": 29643}[workload]
    cmd = [sys.executable] + ((["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                "--master-port", str(port)]) if launcher == "torchrun" else []) + tail
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd="/tmp")
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-1500:]                   # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["dry_run_cpu"] is True and d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["steps"] == 2 and d["warmup"] == 1
    assert d["config"]["workload"] == workload and d["config"]["parallelism"] == "dp2" and d["outputs_finite"] is True
    total_audio = (2 * 3 if workload == "gsr16x10" else 6) * 0.2
    assert abs(d["value"] - total_audio / (d["ms_per_step"] * 1e-3)) < 2e-2 * d["value"]      # whole-job aggregate over both ranks
    if workload == "sharded1024":
        assert d["clips_total"] == 6 and d["gather_matches_direct_restore"] is True
        assert all(d[k] >= 0 for k in ("scatter_ms", "restore_ms", "gather_ms"))
    # a WORLD_SIZE that disagrees with --gpus is refused
    bad = subprocess.run([sys.executable] + tail, capture_output=True, text=True, env=dict(env, WORLD_SIZE="4"), timeout=120, cwd="/tmp")
    assert bad.returncode != 0 and "WORLD_SIZE" in bad.stderr


def test_length_bucketing_in_a_world_of_one():
    """dist.restore_sharded_lengths without a process group = the single-GPU length-bucketing helper: clips of equal length share
    a call (at most max_batch), the results come back in file order."""
    from voicefixer_main_amd import dist as vdist
    calls = []

    def eng(x):
        calls.append(tuple(x.shape))
        return -x
    lens = [30, 10, 30, 30, 10, 7]
    clips = [torch.arange(L, dtype=torch.float32) + i for i, L in enumerate(lens)]
    back = vdist.restore_sharded_lengths(eng, clips, torch.device("cpu"), max_batch=2)
    assert calls == [(2, 30), (1, 30), (2, 10), (1, 7)]
    assert all(torch.equal(b, -c) for b, c in zip(back, clips))
    assert vdist.restore_sharded_lengths(eng, [], torch.device("cpu")) == []


def test_store_hazard_checker_sees_the_pattern(tmp_path):
    """scripts/asm_store_hazard_check.py on hand-written assembly: a 16-byte buffer store with a register soffset followed at once by a
    VALU write of its data registers is reported; with an s_nop in between, with an immediate soffset, with a VALU write of other
    registers, or with an 8-byte store it is not (profiles/r06_store_data_hazard.md)."""
    import subprocess
    head = "\n_ZN3vfx6k_testEv: ; @k\n"
    cases = {
        "hit": ("buffer_store_dwordx4 v[0:3], v9, s[16:19], s79 offen\n v_pk_add_f32 v[0:1], v[4:5], v[60:61]\n", 1),
        "hit_x3": ("buffer_store_dwordx3 v[4:6], v9, s[16:19], s2 offen offset:16\n v_mov_b32_e32 v5, v1\n", 1),
        "nop": ("buffer_store_dwordx4 v[0:3], v9, s[16:19], s79 offen\n s_nop 1\n v_pk_add_f32 v[0:1], v[4:5], v[60:61]\n", 0),
        "imm": ("buffer_store_dwordx4 v[0:3], v9, s[16:19], 0 offen\n v_pk_add_f32 v[0:1], v[4:5], v[60:61]\n", 0),
        "other_regs": ("buffer_store_dwordx4 v[0:3], v9, s[16:19], s79 offen\n v_pk_add_f32 v[4:5], v[4:5], v[60:61]\n", 0),
        "x2": ("buffer_store_dwordx2 v[0:1], v9, s[16:19], s79 offen\n v_mov_b32_e32 v0, v1\n", 0),
    }
    for name, (body, hits) in cases.items():
        f = tmp_path / (name + ".s")
        f.write_text(head + body + " s_endpgm\n")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "asm_store_hazard_check.py"), str(f)], capture_output=True, text=True)
        assert r.returncode == (1 if hits else 0), (name, r.stdout)
        assert ("%d unfenced" % hits) in r.stdout, (name, r.stdout)


def test_no_compiler_touch_of_inflight_weight_registers(tmp_path):
    """The convolution kernels load weight fragments with inline-asm global loads and hand-counted s_waitcnt; a
    compiler-generated copy of such a register before its wait reads stale data (a bug that only shows when the
    register file holds garbage).  Compile both kernels to gfx950 assembly and scan them
    (scripts/asm_inflight_check.py)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "voicefixer_main_amd", "csrc")
    import re
    # kernels that are allowed a few bytes of scratch: opt-in experiments and the opt-in 64-position form of the C = 256 layer
    # (and k_conv<64, ELU, split, ring 3>: 8 bytes since round 1, split-bf16 mode of the vocoder's ELU convolutions only)
    # (and the VL twin of k_conv<64, split>: 8 bytes, the split-bf16 vocoder launches of a varlen batch only -- precision 1)
    may_spill = ("k_convILi64ELb1ELb1ELi0ELi3ELb0ELb0E", "k_convILi64ELb0ELb1ELi0ELi3ELb0ELb0ELb0ELb1E")
    # k_block2d32 (three blocks per CU: 168 registers): its prologue and the request path of BORDER tiles spill (checked below: not the
    # interior tile loop)
    border_spill = ("k_block2d32",)
    for name in ("conv.hip", "resblock.hip", "block2d32.hip", "resblock_w64.hip", "resblock_r128.hip", "resblock_rw.hip", "upsample16.hip",
                 "stft.hip", "small_ops.hip"):
        out = str(tmp_path / (name + ".s"))
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                        "-S", "--cuda-device-only", "-o", out, os.path.join(csrc, name)], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "asm_inflight_check.py"), out],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:]
        # no register spills in the kernels of the default path: a scratch reload is a VMEM operation that queues behind
        # prefetches (the persistent kernels) and costs memory round trips everywhere else
        asm = open(out).read()
        for m in re.finditer(r"\n(_ZN3vfx\w+):.*?; ScratchSize: (\d+)", asm, re.S):
            kernel, scratch = m.group(1), int(m.group(2))
            if any(k in kernel for k in border_spill):
                assert scratch <= 64, (name, kernel, scratch)
            elif not any(k in kernel for k in may_spill):
                assert scratch == 0, (name, kernel, scratch)
            else:
                assert scratch <= 16, (name, kernel, scratch)
        # no FLAT memory instruction anywhere: the compiler's wait-count insertion answers one with lgkmcnt(0) / vmcnt(0) on every later
        # wait, which un-pipelines the fragment reads of kernels whose vmcnt waits are hand-counted asm (conv_common.h: or_flag_global)
        assert not re.search(r"\n\s*flat_(load|store|atomic)", asm), name
        # no 16-byte buffer store with a REGISTER soffset whose data registers the very next VALU instruction overwrites: hipcc 7.2 fences
        # that hazard for immediate soffsets only, gfx950 has it for both (profiles/r06_store_data_hazard.md)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "asm_store_hazard_check.py"), out], capture_output=True, text=True)
        assert r.returncode == 0, (name, r.stdout[-2000:])
        if name == "block2d32.hip":
            # the interior tile loop of k_block2d32 is spill-free: no scratch access between the first and the last MFMA of the loop body
            body = asm[asm.index("v_mfma_f32_32x32x16_bf16"):asm.rindex("v_mfma_f32_32x32x16_bf16")]
            blocks = re.split(r"\n\.LBB\d+_\d+:", body)
            mfma_blocks = [b for b in blocks if b.count("v_mfma") >= 100]
            assert mfma_blocks and all("scratch_" not in b for b in mfma_blocks), "k_block2d32: the convolution blocks spill"


def test_committed_bench_line_follows_the_contract():
    """The committed bench line of the default workload (profiles/r04_bench_gsr16x10.json) carries every key of the
    bench.py contract, including the `roofline`, `cpu_baseline`, in-run `parity`, `step`, `power`, per-step statistics,
    `f32_trunk_mode` and `aux_workloads` objects, with self-consistent numbers; the roofline follows SURVEY.md section 8(d) on the
    fp16 trunk (a ResStack layer = x in + y out = 4 bytes per element)."""
    import json
    d = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_gsr16x10.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "step", "aux_workloads",
              "ms_per_step_min", "ms_per_step_median", "ms_per_step_p90", "step_at_ref_clock", "f32_trunk_mode", "split_bf16_mode"):
        assert k in d, k
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["config"]["workload"] == "gsr16x10" and "model" not in d["config"] and d["config"]["tuning"] == 0
    assert "fp16 residual trunk" in d["dtype"] and "fp32 residual trunk" in d["f32_trunk_mode"]["dtype"]
    assert d["parity"]["logmel_l1"] < d["parity"]["bar"]["logmel_l1"] and d["parity"]["clips"] >= 1 and d["parity"]["wav_sisdr_db"] > 50
    assert d["f16_saturated"] is False and d["negative_input_flag"] == 0 and d["f32_trunk_mode"]["f16_saturated"] is False
    # the K steps of the timed region, one by one: a 3 % box effect can be told from a regression
    assert d["ms_per_step_min"] <= d["ms_per_step_median"] <= d["ms_per_step_p90"] <= d["ms_per_step_max"]
    assert abs(d["ms_per_step_median"] - d["ms_per_step"]) < 0.02 * d["ms_per_step"]
    sr = d["step_at_ref_clock"]
    assert abs(sr["ms"] - d["ms_per_step_median"] * sr["measured_avg_sclk_mhz"] / sr["ref_sclk_mhz"]) < 1e-2
    # the round's A/B inside one line: the fp32 trunk (= round 3's data path) is slower, and a little more accurate
    f32 = d["f32_trunk_mode"]
    assert f32["ms_per_step"] > 1.04 * d["ms_per_step"] and f32["parity"]["wav_sisdr_db"] > d["parity"]["wav_sisdr_db"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "accounting", "hbm_roofline", "mfma_roofline"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # section 8(d): the dominant kernel is a ResStack layer of 16 clips x 49 294 positions x 256 channels on the fp16 trunk: 4 bytes
    # per element -> 768 flop/B, far above the ridge: MFMA-bound; the layout moves exactly that; the counters see the second read
    # of the activated tensor (the residual) miss L2 in part
    hb = r["hbm_roofline"]
    assert "k_resblock<256, 4> f16" in r["kernel"] and r["bound"] == "mfma" and hb["algorithmic_bytes_per_launch"] == 16 * 49294 * 256 * 4
    assert hb["design_bytes_per_launch"] == hb["algorithmic_bytes_per_launch"] and 1.0 < r["traffic"] / hb["design_bytes_per_launch"] < 2.0
    assert r["frac"] > 0.33
    assert abs(hb["achieved"] - hb["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 0.01 * hb["achieved"]
    assert abs(r["achieved"] - r["algorithmic_gflop_per_step"] / r["kernel_ms_per_step"]) < 0.01 * r["achieved"]
    k128 = r["all_conv_kernels"]["k_resblock<128, 4> f16"]
    assert 0.9 < k128["hbm_bytes_per_launch"] / (16 * 147882 * 128 * 4) < 1.2    # C = 128: x (fp16) is read once
    assert r["all_conv_kernels"]["k_resblock_pair<128, 4> f16"]["hbm_bytes_per_launch"] < 1.1 * 16 * 147882 * 128 * 4   # two layers
    for name, k in r["all_conv_kernels"].items():
        assert 0 < k["frac_mfma"] < 1 and 0 < k["frac_hbm"] < 1, name
    st = d["step"]
    assert abs(st["tflops"] - st["gflop"] / st["ms"]) < 0.01 * st["tflops"] and abs(st["frac_of_mfma_peak"] - st["tflops"] / st["peak"]) < 1e-3
    assert abs(st["ms"] - d["ms_per_step"]) < 1e-6
    pw = d["power"]
    assert pw["card_matched_by"] == "pci address" and 500 < pw["avg_power_w"] <= pw["power_cap_w"] and 1000 < pw["avg_sclk_mhz"] <= 2400
    for name in ("ssr_sr64", "stream1s"):
        a = d["aux_workloads"][name]
        for k in ("value", "ms_per_step", "parity", "roofline", "cpu_baseline"):
            assert k in a, (name, k)
        assert "FLOAT64" in a["parity"]["vs"] and a["parity"]["wav_sisdr_db"] > 50
        assert a["cpu_baseline"]["statistic"] == "median" and len(a["cpu_baseline"]["timed_calls"]) >= 3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "cpu_model", "timed_calls"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["statistic"] == "median" and len(c["timed_calls"]) >= 3
    # value = audio seconds of all ranks / wall seconds
    audio = d["n_gpus"] * d["config"]["clips_per_gpu"] * d["config"]["clip_seconds"]
    assert abs(d["value"] - audio / (d["ms_per_step"] * 1e-3)) < 1e-2 * d["value"]
    # the same-box A/B of the round: r03 HEAD and r04 HEAD alternating on one lease
    ab = [json.loads(l) for l in open(os.path.join(ROOT, "profiles", "r04_same_box_ab.jsonl"))]
    r03 = [x["ms_per_step"] for x in ab if x["tree"].startswith("r03")]
    r04 = [x["ms_per_step"] for x in ab if x["tree"].startswith("r04")]
    assert len(r03) >= 2 and len(r04) >= 2 and max(r04) < 0.95 * min(r03)
    assert max(r03) - min(r03) < 0.01 * min(r03) and max(r04) - min(r04) < 0.01 * min(r04)     # same-box repeatability: < 1 %


@pytest.mark.parametrize("C,kind,tuning", [(32, 0, 0), (32, 0, 128), (32, 0, 1024), (64, 0, 0), (32, 1, 0), (32, 2, 0)])
def test_block2d_tiles_cover_every_pixel_once(C, kind, tuning):
    """Tile geometry of the fused 2-D ConvBlockRes (plan_block2d, host-only entry point) -- identity block, entry block (Cin = 1)
    and two-source block -- over image sizes of the shipped shapes and awkward ones: the outputs the kernel's mask lets through
    (restated from resblock.hip: the interior of the h grid, inside the image) hit every pixel exactly once, the patch fits its
    buffer, and the choice among the tiles is the one that computes the fewest h positions."""
    import ctypes
    from voicefixer_main_amd import _lib
    lib = _lib.load_test()
    out = (ctypes.c_int * 8)()
    for H, W in [(1016, 127), (1024, 128), (192, 1024), (128, 127), (7, 5), (1, 1), (14, 14), (15, 29), (508, 63), (3, 200)]:
        assert lib.vfx_plan_block2d_geometry(C, H, W, kind, tuning, out) == 0, (H, W)
        TH, W1, TWo, tiles_h, tiles_w, PW, P, tile_m = list(out)
        MT = 256 if tile_m == 256 else 128
        # (14 x 18 = 252 of the 256 slots: the persistent kernel's second tile, block2d32.hip)
        assert (TH * W1 == MT or (TH, W1, MT) == (14, 18, 256)) and TWo == W1 - 2 and PW == W1 + 2 and P == (TH + 2) * PW and P <= MT + MT // 2
        if kind != 0:
            assert (TH, W1) == (16, 16)                      # the entry / two-source kernels exist on 16 x 16 tiles only
        if C == 64 or tuning & 128:
            assert MT == 128                                 # 16 x 16 tiles: C = 32 without VFX_TUNE_SMALL_2D_TILES
        cands = [(8, 16), (16, 8)] + ([(16, 16)] if (C == 32 and not tuning & 128) else [])
        if C == 32 and kind == 0 and not tuning & (128 | 1024):
            cands.append((14, 18))                           # exists in k_block2d32 only: not under VFX_TUNE_OLD_BLOCK2D
        if kind == 0:
            cost = {c: -(-H // (c[0] - 2)) * -(-W // (c[1] - 2)) * (256 if c[0] * c[1] > 128 else 128) for c in cands}
            assert cost[(TH, W1)] == min(cost.values()), (H, W, cost)
            if (H, W, C) == (1024, 128, 32) and not tuning & (128 | 1024):
                assert (TH, W1) == (14, 18)                  # 128 mel bins: 8 tile columns of 16 instead of 10 of 14
        count = np.zeros((H, W), np.int32)
        for ti in range(tiles_h):
            for tj in range(tiles_w):
                i0, j0 = ti * (TH - 2), tj * TWo
                for li in range(1, TH - 1):
                    for lj in range(1, W1 - 1):
                        r, c = i0 - 1 + li, j0 - 1 + lj
                        if r < H and c < W:
                            count[r, c] += 1
                # the patch pixel of conv1's last tap of the last h pixel, and the two-source block's centre row, stay inside
                assert (TH - 1 + 2) * PW + (W1 - 1 + 2) < P and (TH - 1) * W1 + W1 - 1 < MT
        assert (count == 1).all(), (H, W, TH, W1)
    # the kernels that exist on 16 x 16 tiles only are refused under VFX_TUNE_SMALL_2D_TILES (the builder keeps the two-launch form)
    if kind != 0:
        assert lib.vfx_plan_block2d_geometry(C, 64, 64, kind, 128, out) == 1


@pytest.mark.parametrize("C,precision,tuning", [(64, 2, 0), (64, 1, 0), (128, 2, 0), (256, 2, 0), (256, 2, 64), (64, 2, 8)])
def test_resblock_tiles_cover_every_position_once(C, precision, tuning):
    """Tile geometry of the fused ResStack kernels (plan_resblock, host-only entry point): over the vocoder's dilations, the
    layer pairs of the 16-bit C = 64 stack and short / long / unaligned sequences, the outputs the kernels' masks let through
    -- restated here from resblock.hip / resblock_rw.hip / resblock_w64.hip -- hit every position exactly once, and the taps
    of conv1 stay inside the patch."""
    import ctypes
    from voicefixer_main_amd import _lib
    lib = _lib.load_test()
    out = (ctypes.c_int * 12)()
    cases = [(d, 0) for d in (1, 3, 9, 27, 81, 243, 729, 2187)]
    if C == 64 and precision == 2 and tuning == 0:
        cases += [(1, 3), (9, 27), (3, 9)]
    if C == 128 and precision == 2 and tuning == 0:
        cases += [(1, 3), (2, 4), (16, 1)]
    for T in (3, 90, 1000, 49049, 70001):
        for d, d2 in cases:
            assert lib.vfx_plan_resblock_geometry_tuned(C, T, d, d2, precision, tuning, out) == 0, (T, d, d2)
            fold, TH, W1, TWo, tiles_h, tiles_w, PW, P, MT, rw, patch_rows, asrc = list(out)
            MT = MT or 128
            assert P <= (patch_rows or MT + 64) and TH * W1 <= MT and TH >= 1
            if C == 64 and precision == 2 and tuning == 0 and d2 == 0 and d >= 243 and T > 8 * d:
                assert patch_rows == 384 and TH * TWo >= 240, (d, T, TH, TWo)     # wide folded tiles on the fp16 trunk
            hits = np.zeros(T, dtype=np.int32)
            m = np.arange(MT)
            for ti in range(tiles_h):
                for tj in range(tiles_w):
                    j0 = tj * TWo
                    if d2 > 0:                      # pair: both layers over the index space of the tile
                        pos = j0 - 2 - d2 + m
                        ok = (m >= 2 + d2) & (m <= MT - 3 - d2)
                    else:
                        li, lj = m // W1, m % W1
                        base_h = ti * TH * d + j0 - 1 if fold else j0 - 1
                        pos = base_h + li * (d if fold else 0) + lj
                        ok = (li < TH) & (lj >= 1) & (lj <= W1 - 2)
                        if fold:
                            ok &= j0 + lj - 1 < d
                        # conv1 reads patch rows li * PW + lj + k * (PW if fold else d), k = 0..2
                        rows = np.where(li < TH, li * PW + lj, 0) + 2 * (PW if fold else d)
                        assert rows.max() < P, (T, d, rows.max(), P)
                    ok &= (pos >= 0) & (pos < T)
                    np.add.at(hits, pos[ok], 1)
            assert hits.min() == 1 and hits.max() == 1, (C, precision, T, d, d2, int(hits.min()), int(hits.max()))


def test_wave_fft_pass_structure():
    """The index scheme of the wave-level 1024-point FFT of csrc/stft.hip (wfft1024: 64 lanes x 16 points, Stockham passes of
    radix 16, 16, 4; the 16-point butterfly as n = 4a + b -> k = c + 4d; padded LDS positions i + (i >> 4)), restated in numpy
    with the same per-lane reads, twiddles and writes: it must be the DFT, forward and inverse, and no two lanes of an access may
    collide on a padded position."""
    N = 1024
    rng = np.random.default_rng(0)
    tw = np.exp(-2j * np.pi * np.arange(N) / N)
    zpad = lambda i: i + (i >> 4)

    def fft4(v, sign):
        v0, v1, v2, v3 = v
        a0, a1, a2, d = v0 + v2, v0 - v2, v1 + v3, v1 - v3
        a3 = d * (-1j) if sign < 0 else d * 1j
        return [a0 + a2, a1 + a3, a0 - a2, a1 - a3]

    def fft16(v, sign):
        v = list(v)
        for b in range(4):
            o = fft4([v[b], v[4 + b], v[8 + b], v[12 + b]], sign)
            for c in range(4):
                v[4 * c + b] = o[c] * np.exp(sign * 2j * np.pi * b * c / 16)
        out = [None] * 16
        for c in range(4):
            o = fft4(v[4 * c:4 * c + 4], sign)
            for d in range(4):
                out[c + 4 * d] = o[d]          # the kernel finds X[k] in register 4 (k & 3) + (k >> 2)
        return out

    def wfft(z, sign):
        T = tw if sign < 0 else np.conj(tw)
        lds = np.zeros(N + N // 16 + 1, complex)
        regs = [[z[lane + 64 * r] for r in range(16)] for lane in range(64)]
        used = set()
        for lane in range(64):                  # pass 1: out[16 lane + r] at 17 lane + r
            o = fft16(regs[lane], sign)
            for r in range(16):
                p = 17 * lane + r
                assert p == zpad(16 * lane + r) and p not in used
                used.add(p)
                lds[p] = o[r]
        nxt = np.zeros_like(lds)
        for lane in range(64):                  # pass 2: in[lane + 64 r] at lane + (lane >> 4) + 68 r
            k = lane & 15
            v = []
            for r in range(16):
                p = lane + (lane >> 4) + 68 * r
                assert p == zpad(lane + 64 * r)
                v.append(lds[p] * T[(4 * k * r) % N])
            o = fft16(v, sign)
            for r in range(16):                 # out[16 (lane - k) + k + 16 r] at 17 (lane - k) + k + 17 r
                p = 17 * (lane - k) + k + 17 * r
                assert p == zpad(16 * (lane - k) + k + 16 * r)
                nxt[p] = o[r]
        out = np.zeros(N, complex)
        for lane in range(64):                  # pass 3: four radix-4 butterflies per lane
            for q in range(4):
                j = lane + 64 * q
                v = []
                for r in range(4):
                    p = lane + (lane >> 4) + 68 * q + 272 * r
                    assert p == zpad(j + 256 * r)
                    v.append(nxt[p] * T[(j * r) % N])
                o = fft4(v, sign)
                for r in range(4):
                    out[j + 256 * r] = o[r]
        return out

    z = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    assert np.abs(wfft(z, -1) - np.fft.fft(z)).max() < 1e-10
    assert np.abs(wfft(z, +1) - np.fft.ifft(z) * N).max() < 1e-10
    # frames that run together in the inverse kernel (five rounds, stride D) never share a sample
    hop, nfft = 441, 2048
    D = (nfft + hop - 1) // hop
    assert D * hop >= nfft and D == 5


def test_profile_kernel_names():
    """scripts/kname.py maps the demangled names rocprofv3 records to the names bench.py / libvfx print: the persistent, pair and
    experiment kernels of the ResStack family must land on the `k_resblock<C, NW>` rows of the tables."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from kname import short
    ns = "void vfx::"
    assert short(ns + "k_resblock_rw<8, false, true>(vfx::ResBlockParams const*, int, int)") == "k_resblock<64, 8> f16"
    assert short(ns + "k_resblock_rw<8, true, true>(vfx::ResBlockParams const*, int, int)") == "k_resblock_pair<64, 8> f16"
    assert short(ns + "k_resblock_rw<8, false, false>(vfx::ResBlockParams const*, int, int)") == "k_resblock<64, 8> f16"
    assert short(ns + "k_resblock_w64<256, true>(vfx::ResBlockParams const*)") == "k_resblock<256, 4> f16"
    assert short(ns + "k_resblock_w64<256, false>(vfx::ResBlockParams const*)") == "k_resblock<256, 4> f16"
    assert short(ns + "k_resblock_r128<false, true>(vfx::ResBlockParams const*)") == "k_resblock<128, 4> f16"
    assert short(ns + "k_resblock_r128<true, true>(vfx::ResBlockParams const*)") == "k_resblock_pair<128, 4> f16"
    assert short(ns + "k_resblock_r128<true, false>(vfx::ResBlockParams const*)") == "k_resblock_pair<128, 4> f16"
    assert short(ns + "k_resblock<32, 2, false, true>(vfx::ResBlockParams const*)") == "k_resblock<32, 2>"
    assert short(ns + "k_resblock<64, 4, true, false>(vfx::ResBlockParams const*)") == "k_resblock<64, 4> f16"
    assert short(ns + "k_stft_mel<false>(float const*, int)") == "k_stft_mel<false>"


def test_fragment_reads_of_2d_tiles_are_conflict_free_with_the_2d_key():
    """scripts/lds_conflicts_conv.py models the ds_read_b128 lane groups of gfx950: the 2-D swizzle key k_conv / k_resblock use
    on 2-D tiles must give one LDS cycle per group for every tile shape plan_conv can pick and every tap, and the row-linear key
    (kept for 1-D patches only) must not be better anywhere."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import lds_conflicts_conv as m
    for th, tw in ((8, 16), (16, 8), (24, 4), (32, 4), (8, 2), (64, 2), (4, 32), (2, 64)):
        avg2, worst2 = m.tile_cost(th, tw, m.key_2d)
        avg1, _ = m.tile_cost(th, tw, m.key_1d)
        assert worst2 == 1 and avg2 == 1.0, (th, tw, avg2, worst2)
        assert avg1 >= avg2
    assert m.tile_cost(16, 8, m.key_1d)[0] == 3.0   # what levels 2-4 paid before the BN = 128 tile took the 2-D key
    assert m.tile_cost(1, 128, m.key_2d) == (1.0, 1)  # a 1-D patch: the 2-D key is the row-linear one


def test_upsampler_patches_get_an_even_width():
    """The parity classes of the ResUNets' transposed convolutions read a window ONE wider than the tile (taps 0 / -1): with an odd
    patch width the 2-D swizzle key's premise fails and the model shows 1.5 LDS cycles per lane group (PMC in round 4: 34-37 %
    conflict cycles in those launches).  plan_conv (host-only entry point) must hand such launches an even patch width wherever
    the tile has more than one row, and the modelled fragment reads must then be conflict-free; 3x3 windows are unchanged."""
    import ctypes
    from voicefixer_main_amd import _lib
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import lds_conflicts_conv as m
    lib = _lib.load_test()
    out = (ctypes.c_int * 6)()

    def plan(Hg, Wg, taps):
        dh = (ctypes.c_int * len(taps))(*[t[0] for t in taps])
        dw = (ctypes.c_int * len(taps))(*[t[1] for t in taps])
        assert lib.vfx_plan_conv_geometry(Hg, Wg, len(taps), dh, dw, out) == 0
        return list(out)

    union = [(0, 0), (0, -1), (-1, 0), (-1, -1)]           # what a phased upsampler launch stages (resunet.cpp, upsample)
    for Hg, Wg in [(512, 64), (512, 63), (256, 32), (128, 16), (64, 8), (32, 4), (16, 2), (8, 2), (96, 512), (3, 30), (17, 64)]:
        for taps in (union, [(0, 0), (0, -1)]):             # (the classes without a horizontal tap have a window as wide as the tile)
            TH, TW, PW, P, per_tap, _ = plan(Hg, Wg, taps)
            assert per_tap == 0 and P <= 192
            wide = max(t[1] for t in taps) - min(t[1] for t in taps)
            tall = max(t[0] for t in taps) - min(t[0] for t in taps)
            cols = min(TW, Wg)                                  # (round 6: an image narrower than the tile stages only its own columns)
            if TH > 1 and (TH + tall) * (cols + wide + (cols + wide) % 2) + (TW - cols) <= 192:   # (a 32 x 4 tile has no room for the extra column)
                assert PW % 2 == 0 and PW - cols in (wide, wide + 1), (Hg, Wg, taps, TW, PW)
                offs = [(t[0] - min(u[0] for u in taps), t[1] - min(u[1] for u in taps)) for t in taps]
                assert m.window_cost(TH, TW, PW, offs, m.key_2d) == (1.0, 1), (Hg, Wg, taps, TH, TW, PW)
    TH, TW, PW, P, per_tap, _ = plan(512, 127, [(a, b) for a in (-1, 0, 1) for b in (-1, 0, 1)])
    assert PW == TW + 2 and P == (TH + 2) * PW                      # 3x3: even already
    assert m.window_cost(8, 16, 17, [(0, 0), (0, 1), (1, 0), (1, 1)], m.key_2d)[0] == 1.5   # what the odd width cost


def test_wav_reader_device_path_gives_the_values_of_read(tmp_path, monkeypatch):
    """handlers._WavReader.read_device (round 5): a mono PCM16 file travels as its 2-byte samples and is widened on the device --
    the values are those of `read` / `load_wav` bit for bit (int16 -> float32 -> / 32768 is exact in both places), segment by
    segment incl. the short last one and the read past the end; a stereo file takes the host path (numpy's channel mean)."""
    from voicefixer_main_amd import handlers
    monkeypatch.setattr(torch.Tensor, "pin_memory", lambda self: self)           # no accelerator here: pinning is a no-op
    rng = np.random.default_rng(3)
    x = (rng.normal(size=100001) * 0.3).astype(np.float32)
    mono = str(tmp_path / "m.wav")
    handlers.save_wave(x, mono)
    a, b = handlers._WavReader(mono), handlers._WavReader(mono)
    for n in (60000, 60000, 7):
        want, got = a.read(n), b.read_device(n, "cpu")
        assert got.dtype == torch.float32 and np.array_equal(want, got.numpy())
    assert len(a) == len(b) == 100001
    a.close(), b.close()
    import wave
    st = str(tmp_path / "s.wav")
    with wave.open(st, "wb") as f:
        f.setnchannels(2), f.setsampwidth(2), f.setframerate(44100)
        f.writeframes((rng.integers(-3000, 3000, size=(5000, 2))).astype("<i2").tobytes())
    a, b = handlers._WavReader(st), handlers._WavReader(st)
    assert np.array_equal(a.read(4000), b.read_device(4000, "cpu").numpy()) and np.array_equal(a.read(4000), b.read_device(4000, "cpu").numpy())
    a.close(), b.close()


def test_tall_narrow_images_keep_windowed_stages():
    """The deep ResUNet levels of a LONG clip are tall narrow images (a 60-s segment: 376 x 7, 188 x 3, 94 x 1 pixels).  With
    128 / TW rows per tile no tile shape had an all-taps window inside the 192-row patch buffer and plan_conv fell back to one
    stage per (chunk, tap) -- nine times the stages and no split-K (round 5: 0.13 - 0.36 ms per launch for 0.05 ms of work).
    The planner may now take fewer rows; shapes that fitted before are unchanged (the benched 10-s shapes among them)."""
    import ctypes
    from voicefixer_main_amd import _lib
    lib = _lib.load_test()
    out = (ctypes.c_int * 6)()
    taps = [(a, b) for a in (-1, 0, 1) for b in (-1, 0, 1)]
    dh, dw = (ctypes.c_int * 9)(*[t[0] for t in taps]), (ctypes.c_int * 9)(*[t[1] for t in taps])

    def plan(Hg, Wg):
        assert lib.vfx_plan_conv_geometry(Hg, Wg, 9, dh, dw, out) == 0
        return list(out)
    for Hg, Wg in [(188, 3), (94, 1), (376, 7), (47, 3), (200, 2), (1000, 1)]:
        TH, TW, PW, P, per_tap, tiles = plan(Hg, Wg)
        assert per_tap == 0 and P <= 192 and P == (TH + 2) * PW and TH * TW <= 128 and TH >= 1, (Hg, Wg, list(out))
        assert tiles == -(-Hg // TH) * -(-Wg // TW)
    # the levels of a 10-s clip (Tpad = 1024) and of the 1-s chunk.  Round 6: an image narrower than the tile stages only the columns it
    # has (window width min(TW, Wg) + 2) -- level 6 of a 10-s clip, 32 x 3 pixels, is ONE tile of 32 x 4 (34 x 5 patch pixels) instead of
    # two of 32 x 2: 768 blocks per launch (one round of the chip's slots) instead of 1 536
    assert plan(32, 3) == [32, 4, 5, 170, 0, 1] and plan(16, 1)[:5] == [16, 1, 4, 72, 0]
    assert plan(64, 7)[:2] == [16, 8] and plan(128, 15)[4] == 0 and plan(1024, 127)[4] == 0


def test_tile_split_reciprocal_is_exact_including_one_tile_per_clip():
    """conv_common.h's div_recip(n, r) with the host's r = ceil(2^32 / d) (plan_resblock) -- restated here -- equals n // d for
    every tile count the plan admits (n * d < 2^32), including d = 1 (one tile per clip), where r = 2^32 does not fit 32 bits:
    the first form of this arithmetic kept r in 32 bits and sent every tile of a batch of short sequences to clip 0."""
    def div_recip(n, r):
        return ((n * (r & 0xFFFFFFFF)) >> 32) + (n if (r >> 32) else 0)
    rng = np.random.default_rng(5)
    ds = [1, 2, 3, 7, 126, 127, 792, 4743, 65535, 65536] + [int(x) for x in rng.integers(1, 60000, 200)]
    for d in ds:
        r = ((1 << 32) + d - 1) // d
        assert r >> 33 == 0
        nmax = min((1 << 32) // d - 1, (1 << 31) - 1)
        ns = [0, 1, d - 1, d, d + 1, nmax] + [int(x) for x in rng.integers(0, nmax + 1, 50)]
        for n in ns:
            if 0 <= n <= nmax:
                assert div_recip(n, r) == n // d, (n, d)
    # the 32-bit truncation of r the kernels once used: wrong exactly when d == 1
    assert ((5 * (((1 << 32) // 1) & 0xFFFFFFFF)) >> 32) == 0 != 5 // 1


def test_power_sampler_without_a_gpu_reports_nothing():
    """bench.py's PowerSampler reads amdgpu hwmon files; where there are none (this container) the bench line carries null."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    ps = m.PowerSampler(0)
    with ps:
        pass
    res = ps.result()
    assert res is None or ("avg_power_w" in res and "avg_sclk_mhz" in res)


def test_tuning_bits_agree_between_header_python_and_vfx_create():
    """include/vfx.h's VFX_TUNE_* enum, _lib.py's TUNE_* constants, the names vfx_create announces (api.cpp) and its mask check
    describe the same bits; the GPU suite's TUNING list (tests/test_gpu_models.py) exercises every one of them."""
    import re
    from voicefixer_main_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "vfx.h")).read()
    bits = {m.group(1): int(m.group(2)) for m in re.finditer(r"VFX_TUNE_(\w+)\s*=\s*(\d+)", hdr)}
    assert bits and sorted(bits.values()) == [1 << i for i in range(len(bits))]
    for name, v in bits.items():
        assert getattr(_lib, "TUNE_" + name) == v, name
    api = open(os.path.join(ROOT, "voicefixer_main_amd", "csrc", "api.cpp")).read()
    names = re.search(r"static const char\* names\[\] = \{([^}]*)\}", api).group(1)
    names = re.findall(r'"(\w+)"', names)
    assert names == [n for n, _ in sorted(bits.items(), key=lambda kv: kv[1])]
    assert "tuning & ~%d" % (sum(bits.values())) in api
    gpu = open(os.path.join(ROOT, "tests", "test_gpu_models.py")).read()
    for name, v in bits.items():
        assert '("%s", %d,' % (name, v) in gpu, name
