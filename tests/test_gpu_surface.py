"""GPU: the reference-surface mirror (models / handlers) and HIP-vs-REFERENCE golden vectors."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def voicefixer(engine, unet_sd, voc_sd):
    from voicefixer_main_amd.models import VoiceFixer
    m = VoiceFixer(None, channels=2, type_target="vocals", engine=engine)
    sd = {"generator.analysis_module." + k: v for k, v in unet_sd.items()}
    sd.update({"vocoder.model." + k: v for k, v in voc_sd.items()})
    m.load_state_dict(sd)
    return m.eval().to(torch.device("cuda:0"))


def test_hip_resunet_vs_reference_golden(engine):
    """HIP Generator.forward against the output of the REFERENCE's own module (tests/golden/unet_mel.npz)."""
    g = np.load(os.path.join(G, "unet_mel.npz"))
    got = engine.resunet_mel(torch.from_numpy(g["mel_in"][:, 0])).cpu().numpy()
    d = np.abs(got - g["logmel_out"][:, 0])
    assert d.mean() < engine.tol['logmel_l1'] and d.max() < engine.tol['logmel_max'], (d.mean(), d.max())  # bar: L1 <= 1e-3


def test_hip_mel_filterbank_vs_reference_golden(voicefixer):
    g = np.load(os.path.join(G, "mel_fb.npz"))
    fb = np.zeros((1025, 128), np.float32)
    fb[g["rows"], g["cols"]] = g["vals"]
    sp = np.abs(np.random.default_rng(1).normal(size=(1, 1, 5, 1025))).astype(np.float32)
    got = voicefixer.mel(torch.from_numpy(sp).cuda().permute(0, 1, 3, 2)).permute(0, 1, 3, 2).cpu().numpy()
    assert np.abs(got - sp.astype(np.float64) @ fb.astype(np.float64)).max() < 2e-5 * got.max()


def test_voicefixer_surface_matches_oracle(voicefixer, unet_sd, voc_sd):
    from oracle import dsp, resunet
    from oracle import vocoder as ovoc
    from voicefixer_main_amd import synth
    from voicefixer_main_amd.models import from_log
    wav = synth.make_clips(1, 0.7, seed=5)
    x = torch.from_numpy(wav).cuda()
    sp, cos, sin = voicefixer.f_helper.wav_to_spectrogram_phase(x)
    assert sp.shape == cos.shape == sin.shape == (1, 1, wav.shape[-1] // 441 + 1, 1025)
    sp2, mel = voicefixer.pre(x)
    ref_sp, ref_mel = dsp.wav_to_mel(wav.astype(np.float64))
    assert np.abs(mel.cpu().numpy() - ref_mel).max() < 1e-5 * ref_mel.max()
    out = voicefixer(mel)["mel"]
    ref_log = resunet.generator_mel(unet_sd, torch.from_numpy(ref_mel.astype(np.float32))).numpy()
    assert np.abs(out.cpu().numpy() - ref_log).mean() < voicefixer.engine.tol['logmel_l1']
    wave = voicefixer.vocoder(from_log(out))
    ref_wave = ovoc.vocoder(voc_sd, torch.from_numpy(dsp.from_log(ref_log))).numpy()
    assert wave.shape == ref_wave.shape
    assert np.abs(wave.cpu().numpy() - ref_wave).max() < 1e-3
    # fused front-end == unfused
    fused = voicefixer.f_helper.wav_to_mel(x)
    assert np.abs(fused.cpu().numpy() - mel.cpu().numpy()).max() < 1e-6 * ref_mel.max()
    bad = mel.clone()
    bad[0, 0, 0, 0] = -1.0
    with pytest.raises(AssertionError):
        voicefixer(bad)


def test_ssr_unet_vs_reference_golden_and_oracle(engine):
    """Spectrogram ResUNet + phase recombination + ISTFT (unet_v2.py:86-148)."""
    from oracle import pipeline
    from voicefixer_main_amd import synth
    from voicefixer_main_amd.models import SSR_UNet
    g = np.load(os.path.join(G, "unet_spec.npz"))
    sd = synth.make_resunet_state_dict(2)
    m = SSR_UNet(None, channels=1, engine=engine)
    m.load_state_dict({"generator.unet." + k: v for k, v in sd.items()})
    wav = torch.from_numpy(g["wav_in"]).cuda()
    sp, _ = m.pre(wav)
    out = m(sp, wav)
    assert out["wav"].shape == wav.shape and out["clean"] is sp
    got = out["wav"].cpu().numpy()
    scale = max(1.0, np.abs(g["wav_out"]).max())
    assert np.abs(got - g["wav_out"]).max() < (2e-4 if engine.tol['name'] == 'fp32' else 3e-3) * scale    # vs the reference trunk
    ref = pipeline.restore_ssr(sd, g["wav_in"])
    err = got - ref["wav"]
    sisdr = 10 * np.log10((ref["wav"] ** 2).sum() / ((err ** 2).sum() + 1e-30))
    assert sisdr > (70.0 if engine.tol['name'] == 'fp32' else 45.0), sisdr


def test_ssr_unet_vs_reference_golden_batch_of_two(engine):
    """The spectrogram path against the REFERENCE module's own output on B = 2 full-band clips of T = 130 frames (Tpad = 192,
    unaligned length; tests/golden/unet_spec_b2.npz).  On full-band input the comparison is not limited by the phase of empty
    bins (two fp32 evaluations agree to > 90 dB, tests/test_oracle_golden.py), so these bars are accuracy statements for the
    kernels: >= 88 dB with split-bf16 operands, >= 105 dB in the exact-fp32 mode (measured on MI355X: 95.2 / 110.8-113.1 dB,
    profiles/r04_ssr_golden_b2.txt)."""
    from voicefixer_main_amd import synth
    from voicefixer_main_amd.models import SSR_UNet
    g = np.load(os.path.join(G, "unet_spec_b2.npz"))
    m = SSR_UNet(None, channels=1, engine=engine)
    m.load_state_dict({"generator.unet." + k: v for k, v in synth.make_resunet_state_dict(2).items()})
    wav = torch.from_numpy(g["pcm_in"].astype(np.float32) / 32768.0)[:, None].cuda()
    sp, _ = m.pre(wav)
    assert sp.shape == (2, 1, 130, 1025)
    got = m(sp, wav)["wav"][:, 0].cpu().numpy().astype(np.float64)
    ref = g["wav_out"].astype(np.float64)
    assert got.shape == ref.shape
    for b in range(2):                          # per clip: clip 1 must not borrow anything from clip 0
        s = 10 * np.log10((ref[b] ** 2).sum() / (((got[b] - ref[b]) ** 2).sum() + 1e-30))
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/ssr_golden_b2.txt", "a") as f:
            f.write("%s clip %d: %.2f dB\n" % (engine.tol['name'], b, s))
        assert s > (105.0 if engine.tol['name'] == 'fp32' else 88.0), (b, s)
    tail = ref.shape[1] % 441
    rel = np.abs(got[:, -tail:] - ref[:, -tail:]).max() / np.abs(ref[:, -tail:]).max()
    assert rel < (1e-4 if engine.tol['name'] == 'fp32' else 1e-3), rel


def test_restore_list_of_unequal_lengths_equals_one_call_per_clip(voicefixer):
    """VoiceFixer.restore_list (dist.restore_sharded_lengths in a world of one): five clips of three lengths -- equal lengths run
    as one batch, longest first -- give, clip by clip, EXACTLY what a batch-of-one `restore` of that clip gives (the reference
    calls its handler once per file), in the order of the input list."""
    from voicefixer_main_amd import synth
    lens = [30000, 12345, 30000, 20001, 12345]
    clips = [torch.from_numpy(synth.make_clips(1, L / 44100.0, seed=50 + i)[0, 0]).cuda() for i, L in enumerate(lens)]
    assert [int(c.shape[0]) for c in clips] == lens
    got = voicefixer.restore_list(clips)
    assert [int(g.shape[0]) for g in got] == lens
    for c, g in zip(clips, got):
        one = voicefixer.restore(c[None])[0]
        assert torch.isfinite(g).all() and torch.equal(g, one)


VARLEN = [28500, 56000, 33333, 50017, 41000]      # T = 65, 127, 76, 114, 93 frames: one padded length (128), odd and even T


def _varlen_batch(lens, seed=70):
    from voicefixer_main_amd import synth
    clips = [torch.from_numpy(synth.make_clips(1, L / 44100.0 + 0.01, seed=seed + i)[0, 0, :L].copy()).cuda() for i, L in enumerate(lens)]
    x = torch.full((len(lens), max(lens)), 0.37, device="cuda")          # the padding is NOT silence: it must not matter
    for j, c in enumerate(clips):
        x[j, :lens[j]] = c
    return clips, x


@pytest.mark.parametrize("unify", [False, True])
def test_restore_varlen_equals_one_call_per_clip_and_the_oracle(engine, unet_sd, voc_sd, unify):
    """vfx_restore_gsr_varlen: five clips of five lengths as ONE padded batch give, clip by clip, what a batch-of-one
    vfx_restore_gsr of that clip gives -- bit for bit: the tile a position falls into differs, the sums that make its value do
    not (with unify_energy: to the last bits of a float atomicAdd sum) -- and what the oracle computes for the clip on its own;
    past a clip's end both outputs are zero."""
    from oracle import pipeline
    clips, x = _varlen_batch(VARLEN)
    out, logmel = engine.restore_gsr_varlen(x, VARLEN, unify_energy=unify, want_logmel=True)
    assert engine.take_flags() & 3 == 0
    tol = engine.tol
    for j, (c, L) in enumerate(zip(clips, VARLEN)):
        one, lm1 = engine.restore_gsr(c[None], unify_energy=unify, want_logmel=True)
        T = L // 441 + 1
        if unify:    # the energies of amp_to_original_f are float atomicAdd sums: their order, hence the last bits of the scale,
            #          differ from launch to launch -- also between two identical calls of either entry point
            #          (the 16-bit vocoder turns a last-bit change of its input into fp16 rounding steps: its own waveform bar)
            assert float((out[j, :L] - one[0]).abs().max()) < max(2e-5, tol["voc_max"]), j
        else:
            assert torch.equal(out[j, :L], one[0]), (j, float((out[j, :L] - one[0]).abs().max()))
        assert torch.equal(logmel[j, :T], lm1[0]), j
        assert float(out[j, L:].abs().max()) == 0.0 if L < x.shape[1] else True
        assert float(logmel[j, T:].abs().max()) == 0.0 if T < logmel.shape[1] else True
    for j in (0, 1):          # the shortest and the longest clip against the CPU oracle, each on its own
        L = VARLEN[j]
        ref = pipeline.restore_gsr(unet_sd, voc_sd, clips[j].cpu().numpy()[None, None], unify_energy=unify)
        lm = logmel[j, :L // 441 + 1].cpu().numpy()
        assert np.abs(lm - ref["logmel"][0, 0]).mean() < tol["logmel_l1"]
        err = out[j, :L].cpu().numpy().astype(np.float64) - ref["wav"][0, 0]
        sisdr = 10 * np.log10((ref["wav"].astype(np.float64) ** 2).sum() / ((err ** 2).sum() + 1e-30))
        assert sisdr > tol["sisdr"], (j, sisdr)


def test_restore_two_lengths_with_one_frame_count(engine, unet_sd, voc_sd):
    """Two clips whose lengths differ but whose frame counts agree (L // 441 + 1 = 30), one after the other on ONE handle: each
    is restored with its OWN length (rounds 1-4 cached the fused plan by frame count: the second call ran with the first
    one's sample count -- wrong reflection point, wrong trim)."""
    from oracle import pipeline
    from voicefixer_main_amd import synth
    for L in (13000, 12800, 13100):
        wav = synth.make_clips(1, L / 44100.0 + 0.01, seed=31)[..., :L]
        ref = pipeline.restore_gsr(unet_sd, voc_sd, wav)["wav"][:, 0]
        out = engine.restore_gsr(torch.from_numpy(wav[:, 0])).cpu().numpy().astype(np.float64)
        err = out - ref
        sisdr = 10 * np.log10((ref.astype(np.float64) ** 2).sum() / ((err ** 2).sum() + 1e-30))
        assert sisdr > engine.tol["sisdr"], (L, sisdr)


def test_ssr_restore_list_of_unequal_lengths(engine):
    """The spectrogram-domain twin (vfx_restore_ssr_varlen through SSR_UNet.restore_list): four clips of four lengths in one
    padded-frame bucket plus one in another give, clip by clip, what `pre` + `forward` of that clip alone give -- bit for bit --
    and what the float64 oracle computes for it (the full-band bars of the spectrogram path)."""
    from oracle import pipeline
    from voicefixer_main_amd import synth
    from voicefixer_main_amd.models import SSR_UNet
    sd = synth.make_resunet_state_dict(2)
    m = SSR_UNet(None, channels=1, engine=engine)
    m.load_state_dict({"generator.unet." + k: v for k, v in sd.items()})
    lens = [28500, 41000, 33333, 50017, 14312]                    # frames 65, 93, 76, 114 (pad to 128) and 33 (pads to 64)
    clips, _ = _varlen_batch(lens, seed=130)
    got = m.restore_list(clips)
    assert [int(g.shape[0]) for g in got] == lens
    for c, g in zip(clips, got):
        x = c[None, None]
        sp, _ = m.pre(x)
        assert torch.equal(g, m(sp, x)["wav"][0, 0])
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    for j in (0, 3):
        ref = pipeline.restore_ssr(sd64, clips[j].cpu().numpy().astype(np.float64)[None, None], dtype=torch.float64)["wav"][0, 0]
        err = got[j].cpu().numpy().astype(np.float64) - ref
        s = 10 * np.log10((ref ** 2).sum() / ((err ** 2).sum() + 1e-30))
        assert s > (105.0 if engine.tol['name'] == 'fp32' else 88.0), (j, s)


def test_restore_varlen_sub_batches_and_errors(engine, monkeypatch):
    clips, x = _varlen_batch(VARLEN, seed=90)
    want = engine.restore_gsr_varlen(x, VARLEN)
    monkeypatch.setenv("VFX_MAX_CLIPS", "2")          # three launches of the plan: 2 + 2 + 1 clips
    got = engine.restore_gsr_varlen(x, VARLEN)
    monkeypatch.delenv("VFX_MAX_CLIPS")
    assert torch.equal(got, want)
    # clips of DIFFERENT padded frame counts in one call (round 6): 20000 samples = 46 frames pad to 64, the others to 128 -- the
    # ResUNet runs per padded count inside the call, the vocoder once; every clip still equals its own call
    mixed = [20000] + VARLEN[1:]
    got = engine.restore_gsr_varlen(x, mixed)
    for j, L in enumerate(mixed):
        assert torch.equal(got[j, :L], engine.restore_gsr(x[j:j + 1, :L].contiguous())[0]), j
        assert not bool(got[j, L:].any())
    with pytest.raises(RuntimeError, match="samples"):
        engine.restore_gsr_varlen(x, [900] + VARLEN[1:])
    with pytest.raises(RuntimeError, match="samples"):
        engine.restore_gsr_varlen(x, [x.shape[1] + 1] + VARLEN[1:])
    engine.take_flags()


def test_two_handles_on_two_streams_give_the_sequential_results(engine):
    """Two handles of one device driven from two HIP streams, batches alternating between them with nothing waited for in between:
    every output equals, bit for bit, what the first handle alone computes on one stream.  (Round 5: with the launches of the two
    streams free to overlap, k_voc_final's sums went wrong in lanes 48-63 of single instructions whenever the other stream's 16-bit
    MFMA convolutions ran at the same time -- a few hundred samples per batch, 1e-4 .. 1e-2 off, in the 16-bit AND the split-bf16 mode;
    profiles/r05_two_streams.md.  StreamTurn (csrc/vfx_internal.h) now makes calls on different streams take turns.)"""
    from tests.conftest import _make_engine
    from voicefixer_main_amd import synth
    # two FRESH handles: the session's `engine` carries state other tests gave it -- the `voicefixer` fixture's MelScale registers
    # the torch-evaluated mel filterbank (bit-identical to the reference's buffer) in place of the library's built-in table, a
    # last-bit difference that reaches every sample and has nothing to do with streams
    engine, twin = _make_engine(engine.cfg.precision), _make_engine(engine.cfg.precision)
    base = torch.from_numpy(synth.make_clips(6, 3.0, seed=31)[:, 0]).cuda()
    wavs = [base[:, :60000 + 12000 * k].contiguous() for k in range(6)]
    ref = [engine.restore_gsr(w) for w in wavs]
    mels = [engine.stft(w)["mel"] for w in wavs]
    ref_voc = [engine.vocoder(m) for m in mels]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    engines = [engine, twin]
    for rep in range(3):
        got, got_voc = [], []
        torch.cuda.synchronize()
        for i, w in enumerate(wavs):
            with torch.cuda.stream(streams[i % 2]):
                got.append(engines[i % 2].restore_gsr(w))
            with torch.cuda.stream(streams[(i + 1) % 2]):       # the vocoder alone beside the other handle's full restore
                got_voc.append(engines[(i + 1) % 2].vocoder(mels[i]))
        torch.cuda.synchronize()
        for i in range(len(wavs)):
            assert torch.equal(got[i], ref[i]), (rep, i, int((got[i] != ref[i]).sum()))
            assert torch.equal(got_voc[i], ref_voc[i]), (rep, i, int((got_voc[i] != ref_voc[i]).sum()))
    assert engine.take_flags() & 3 == 0 and twin.take_flags() & 3 == 0


def test_graph_replay_takes_its_turn_beside_a_live_call_on_another_stream(engine):
    """A hipGraph captured from one handle and replayed on stream A through Engine.replay (vfx_turn_begin / vfx_turn_end around the
    replay) while a second handle makes live calls on stream B: every output equals the sequential one bit for bit.  A captured
    call is exempt from the library's turns, so a bare graph.replay() beside another stream's calls would be exactly the overlap
    round 5 found faulty ("Streams"); the bracket restores the rule."""
    from tests.conftest import _make_engine
    from voicefixer_main_amd import synth
    e1, e2 = _make_engine(engine.cfg.precision), _make_engine(engine.cfg.precision)
    base = torch.from_numpy(synth.make_clips(6, 3.0, seed=37)[:, 0]).cuda()
    wav_g = base[:, :90000].contiguous()
    live = [base[:, :60000 + 12000 * k].contiguous() for k in range(4)]
    ref_g = e1.restore_gsr(wav_g).clone()
    ref_live = [e2.restore_gsr(w).clone() for w in live]
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    out_g = torch.empty_like(wav_g)
    graph = torch.cuda.CUDAGraph()
    sa.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(sa):
        e1.restore_gsr(wav_g, out=out_g)
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=sa):
            e1.restore_gsr(wav_g, out=out_g)
    torch.cuda.synchronize()
    for rep in range(3):
        got = []
        for w in live:
            with torch.cuda.stream(sa):
                out_g.zero_()
                e1.replay(graph)
                g_now = out_g.clone()
            with torch.cuda.stream(sb):
                got.append(e2.restore_gsr(w))
            with torch.cuda.stream(sa):
                assert torch.equal(g_now, ref_g), (rep, int((g_now != ref_g).sum()))    # (synchronises stream A only)
        torch.cuda.synchronize()
        for i, w in enumerate(live):
            assert torch.equal(got[i], ref_live[i]), (rep, i, int((got[i] != ref_live[i]).sum()))
    assert e1.take_flags() & 3 == 0 and e2.take_flags() & 3 == 0
    del graph
    e1.unpin_plans()


def test_restore_list_buckets_by_padded_frames(voicefixer):
    """restore_list on clips of seven lengths with two padded frame counts: ONE call of the library instead of seven (round 6: the
    ResUNet per padded count inside the call, one vocoder pass), results equal to one `restore` per clip."""
    lens = [30000, 12345, 41000, 20001, 12345, 28224, 50017]       # T = 69, 28, 93, 46, 28, 65, 114 -> padded 128, 64, 128, 64, 64, 128, 128
    clips, _ = _varlen_batch(lens, seed=110)
    calls = []
    eng = voicefixer.engine
    orig_v, orig_f = eng.restore_gsr_varlen, eng.restore_gsr
    eng.restore_gsr_varlen = lambda x, l, **k: (calls.append(("varlen", x.shape[0])), orig_v(x, l, **k))[1]
    eng.restore_gsr = lambda x, **k: (calls.append(("fixed", x.shape[0])), orig_f(x, **k))[1]
    try:
        got = voicefixer.restore_list(clips)
    finally:
        del eng.restore_gsr_varlen, eng.restore_gsr
    assert sorted(calls) == [("varlen", 7)], calls
    for c, g in zip(clips, got):
        assert torch.equal(g, voicefixer.restore(c[None])[0])


def test_handler_end_to_end(voicefixer, unet_sd, voc_sd, tmp_path):
    from oracle import pipeline
    from voicefixer_main_amd import handlers, synth
    wav = synth.make_clips(1, 1.3, seed=21)
    src, dst = str(tmp_path / "in.wav"), str(tmp_path / "out.wav")
    handlers.save_wave(wav[0, 0], src)
    handlers._state["model"] = voicefixer
    metrics = handlers.handler(src, dst, src, ckpt=None, device=torch.device("cuda:0"), needrefresh=False,
                               meta={"unify_energy": False})
    assert set(metrics) == {"mel-lsd", "mel-sispec", "mel-non-log-sispec", "mel-ssim"}     # eval_gsr_voicefixer.py:59-64
    assert all(isinstance(v, float) and np.isfinite(v) for v in metrics.values()) and -1.0 <= metrics["mel-ssim"] <= 1.0
    # a target of another length is an error, as in the reference (no silent trimming)
    short = str(tmp_path / "short.wav")
    handlers.save_wave(wav[0, 0, :30000], short)
    with pytest.raises(RuntimeError):
        handlers.handler(src, dst, short, ckpt=None, device=torch.device("cuda:0"), needrefresh=False, meta={"unify_energy": False})
    out = handlers.load_wav(dst)
    x = handlers.load_wav(src)
    assert out.shape == x.shape
    ref = pipeline.restore_gsr(unet_sd, voc_sd, x[None, None])["wav"][0, 0]
    assert np.abs(out - ref).max() < 1e-3 + 2.0 / 32768


def _handler_in_reference_order(model, src, dst):
    """eval_gsr_voicefixer.py:37-77 exactly as written -- whole-file load, per-segment calls with their host syncs
    (to_log's assert inside model(), `if torch.max(torch.abs(out)) > 1.0`), torch.cat, save_wave of the whole file."""
    from voicefixer_main_amd import handlers
    from voicefixer_main_amd.models import from_log, tensor2numpy
    dev = torch.device("cuda:0")
    wav_10k = handlers.load_wav(src, sample_rate=44100)
    res = []
    seg_length = 44100 * handlers.SEG_SECONDS
    break_point = seg_length
    while break_point < wav_10k.shape[0] + seg_length:
        segment = wav_10k[break_point - seg_length:break_point]
        _, mel_noisy, seg_t = handlers._pre(model, segment, dev)
        out_model = model(mel_noisy)
        denoised_mel = from_log(out_model["mel"])
        out = model.vocoder(denoised_mel)
        if torch.max(torch.abs(out)) > 1.0:
            out = out / torch.max(torch.abs(out))
        out, _ = handlers.trim_center(out, seg_t)
        res.append(out)
        break_point += seg_length
    out = torch.cat(res, -1)
    handlers.save_wave(tensor2numpy(out[0, ...]), fname=dst, sample_rate=44100)


def test_handler_streams_segments_but_writes_the_same_file(voicefixer, tmp_path, monkeypatch, capsys):
    """The shipped handler reads, restores and writes a file segment by segment with ONE host sync per file (flags and
    peaks checked at the end, PCM conversion on the device); its output file must be byte-identical to the reference
    control flow with its two syncs per segment.  (The TFGAN generator ends in tanh, so |out| <= 1: the peak-normalising
    branch and its warning cannot fire on this vocoder; the device-side form of it is value-identical by construction.)"""
    from voicefixer_main_amd import handlers, synth
    monkeypatch.setattr(handlers, "SEG_SECONDS", 1)          # 3 segments (the last one short) out of a 2.4-s file
    wav = synth.make_clips(1, 2.4, seed=33)
    src, a, b = str(tmp_path / "in.wav"), str(tmp_path / "a.wav"), str(tmp_path / "b.wav")
    handlers.save_wave(wav[0, 0], src)
    handlers._state["model"] = voicefixer
    dev = torch.device("cuda:0")
    handlers.handler(src, a, None, ckpt=None, device=dev, needrefresh=False, meta={"unify_energy": False})
    _handler_in_reference_order(voicefixer, src, b)
    assert open(a, "rb").read() == open(b, "rb").read()
    assert handlers.load_wav(a).shape == (wav.shape[-1],)
    assert "Exceed energy limit" not in capsys.readouterr().out
    assert voicefixer.engine.take_flags() == 0


def test_unify_energy_path(engine, unet_sd, voc_sd):
    from oracle import pipeline
    from voicefixer_main_amd import synth
    wav = synth.make_clips(2, 0.6, seed=31, mode="lowpass")
    ref = pipeline.restore_gsr(unet_sd, voc_sd, wav, unify_energy=True)
    out = engine.restore_gsr(torch.from_numpy(wav[:, 0]), unify_energy=True).cpu().numpy()
    err = out - ref["wav"][:, 0]
    sisdr = 10 * np.log10((ref["wav"] ** 2).sum() / ((err ** 2).sum() + 1e-30))
    assert sisdr > 50.0, sisdr


def test_graph_capture_restore_and_streaming_step(engine):
    """BASELINE configs[4]: a steady-state call enqueues kernels only (no allocation, no sync), so a whole
    restore step -- and the gsr_unet 1-s streaming step -- can be captured into a hipGraph and replayed."""
    from voicefixer_main_amd import synth
    from voicefixer_main_amd.engine import MODEL_UNET_SPEC
    wav = torch.from_numpy(synth.make_clips(2, 1.0, seed=41)[:, 0]).cuda()
    out_eager = engine.restore_gsr(wav).clone()          # warm-up: builds the plan, sizes the arena, sets kernel attributes
    out = torch.empty_like(wav)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        engine.restore_gsr(wav, out=out)                  # second warm-up on the capture stream
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            engine.restore_gsr(wav, out=out)
    torch.cuda.current_stream().wait_stream(side)
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, out_eager)
    # streaming step: spectrogram ResUNet on one 1-s chunk (T = 101 -> Tpad = 128)
    engine.load_state_dict(MODEL_UNET_SPEC, synth.make_resunet_state_dict(2))
    chunk = wav[:1].contiguous()
    sp = engine.stft(chunk, want_mel=False, want_sp=True)["sp"]
    ref = engine.resunet_spec(sp, chunk).clone()
    g2 = torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream())  # one handle = one stream at a time: the arena is shared state
    with torch.cuda.stream(side):
        y = engine.resunet_spec(sp, chunk)
        torch.cuda.synchronize()
        with torch.cuda.graph(g2, stream=side):
            y = engine.resunet_spec(sp, chunk)
    torch.cuda.current_stream().wait_stream(side)
    y.zero_()
    g2.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, ref)
    # A handle with a captured plan refuses to MOVE its arena (the graph's kernels hold absolute pointers into it): a call that
    # needs a larger workspace fails loudly instead of freeing memory the graph still replays into -- until the caller declares the
    # graphs destroyed (vfx_unpin_plans).  A fresh engine, so that the session-wide one keeps its arena.
    from voicefixer_main_amd.engine import Engine
    e2 = Engine("cuda:0", config={"precision": engine.precision})
    e2.load_state_dict(MODEL_UNET_SPEC, synth.make_resunet_state_dict(2))
    small = wav[:1, :22050].contiguous()
    sps = e2.stft(small, want_mel=False, want_sp=True)["sp"]
    e2.resunet_spec(sps, small)
    g3 = torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        torch.cuda.synchronize()
        with torch.cuda.graph(g3, stream=side):
            ys = e2.resunet_spec(sps, small)
    torch.cuda.current_stream().wait_stream(side)
    with pytest.raises(RuntimeError, match="vfx_reserve"):
        e2.resunet_spec(sp, chunk)                       # twice the frames: the arena would have to grow
    g3.replay()
    torch.cuda.synchronize()
    assert torch.isfinite(ys).all()
    del g3
    e2.unpin_plans()
    assert torch.equal(e2.resunet_spec(sp, chunk), ref)  # now it may grow
    del graph, g2
    engine.unpin_plans()                                 # the session-wide engine may move its arena again in later tests


def _toy_net(x):
    """oracle.chunker.toy_nnet as device code (test stand-in for a network)."""
    prev = torch.cat([torch.zeros_like(x[..., :1]), x[..., :-1]], -1)
    n = x.shape[-1]
    ramp = (torch.arange(n, dtype=torch.float32, device=x.device) / n) * 0.05
    return {"wav": 0.6 * x + 0.3 * prev + ramp}


def test_chunkers_vs_reference_golden(engine):
    """Both long-audio chunkers (batched nnet calls + vfx_chunk_gather / vfx_chunk_ola) against the outputs of
    the REFERENCE classes (tests/golden/chunker.npz), and against the oracle at an unaligned length."""
    from oracle import chunker as oc
    from voicefixer_main_amd.chunker import LambdaOverlapAdd, LambdaOverlapAddBoxcar
    g = np.load(os.path.join(G, "chunker.npz"))
    names = [k[:-2] for k in g.files if k.endswith("_y")]
    assert len(names) == 10
    for name in names:
        x, y = g[name + "_x"], g[name + "_y"]
        W, p, hann = [int(v) for v in g[name + "_cfg"]]
        w = "hanning" if hann else "boxcar"
        if name.startswith("ola"):
            m = LambdaOverlapAdd(_toy_net, 1, W, hop_size=p, window=w, engine=engine, max_batch=5)
        else:
            m = LambdaOverlapAddBoxcar(_toy_net, 1, W, p, window=w, reorder_chunks=False, engine=engine, max_batch=3)
        got = m(torch.from_numpy(x).cuda()).cpu().numpy()
        assert got.shape == y.shape, name
        assert np.abs(got - y).max() <= 1e-6, (name, np.abs(got - y).max())
    # odd sizes (scalar gather path), un-windowed branch
    x = np.random.default_rng(9).uniform(-1, 1, (2, 1, 1237)).astype(np.float32)
    m = LambdaOverlapAdd(_toy_net, 1, 126, hop_size=21, window=None, engine=engine)
    ref = oc.overlap_add(oc.toy_nnet, x, 126, 21, None)
    assert np.abs(m(torch.from_numpy(x).cuda()).cpu().numpy() - ref).max() <= 2e-6
    m = LambdaOverlapAddBoxcar(_toy_net, 1, 126, 7, window=None, engine=engine)
    ref = oc.overlap_add_boxcar(oc.toy_nnet, x, 126, 7, None)
    assert np.abs(m(torch.from_numpy(x).cuda()).cpu().numpy() - ref).max() <= 1e-6
    with pytest.raises(NotImplementedError):
        LambdaOverlapAdd(_toy_net, 2, 64, engine=engine)


def test_chunked_restore_batched_equals_sequential(voicefixer):
    """A 3.3-s clip restored through 1-s frames with 0.1-s margins: all inner frames as one batch must be
    bit-identical to the reference's one-chunk-at-a-time order, and every frame equal to restoring that frame
    (with its margins) on its own."""
    from voicefixer_main_amd import synth
    from voicefixer_main_amd.chunker import LambdaOverlapAdd, LambdaOverlapAddBoxcar
    wav = torch.from_numpy(synth.make_clips(1, 3.3, seed=21)).cuda()            # (1, 1, L)
    nnet = lambda c: {"wav": voicefixer.restore(c)}
    W, M = 44100, 4410
    batched = LambdaOverlapAddBoxcar(nnet, 1, W, M, window=None, engine=voicefixer.engine)(wav)
    seq = LambdaOverlapAddBoxcar(nnet, 1, W, M, window=None, engine=voicefixer.engine, max_batch=1)(wav)
    assert batched.shape == wav.shape and torch.isfinite(batched).all()
    assert torch.equal(batched, seq)
    own = voicefixer.restore(wav[..., W - M:2 * W + M].contiguous())[..., M:-M]   # frame 1 on its own
    assert torch.equal(batched[..., W:2 * W], own)
    a = LambdaOverlapAdd(nnet, 1, W, window="hanning", engine=voicefixer.engine)(wav)
    b = LambdaOverlapAdd(nnet, 1, W, window="hanning", engine=voicefixer.engine, max_batch=1)(wav)
    assert a.shape == wav.shape and torch.equal(a, b)


def test_spectral_metrics_kernel_vs_oracle_and_reference(engine):
    """vfx_spectral_metrics (per-clip LSD and SiSpec in one pass, double-precision inner products) against the float64
    oracle and the outputs of the reference's own metric code (tests/golden/metrics.npz)."""
    from oracle import metrics as om
    g = np.load(os.path.join(G, "metrics.npz"))
    for tag in "abc":
        e, t = g[tag + "_est"], g[tag + "_tgt"]
        got = engine.spectral_metrics(torch.from_numpy(e), torch.from_numpy(t)).cpu().numpy()
        assert got.shape == (e.shape[0], 2)
        assert np.abs(got[:, 0] - om.lsd(e, t)[:, 0]).max() < 2e-6 * max(1.0, got[:, 0].max())
        assert np.abs(got[:, 1] - om.sispec_per_clip(e, t)).max() < 1e-3
        assert np.abs(got[:, 0] - g[tag + "_lsd"][:, 0, 0, 0]).max() < 2e-6 * max(1.0, got[:, 0].max())
        assert abs(got[:, 1].mean() - float(g[tag + "_sispec_lin"])) < 2e-3
        le, lt = np.log10(np.clip(e, 1e-8, None)), np.log10(np.clip(t, 1e-8, None))
        got = engine.spectral_metrics(torch.from_numpy(le), torch.from_numpy(lt)).cpu().numpy()
        assert abs(got[:, 1].mean() - float(g[tag + "_sispec_log"])) < 2e-3
    x = torch.rand(2, 1, 9, 128) + 0.1
    m = engine.spectral_metrics(x, x).cpu().numpy()
    assert m[:, 0].max() < 1e-6 and m[:, 1].min() > 100.0      # identical pair: LSD 0, SiSpec at the eps ceiling
