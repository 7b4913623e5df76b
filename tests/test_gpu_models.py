"""GPU parity of the model stages / whole path against the CPU oracle, through the C ABI."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# Tolerances come from conftest.TOL per arithmetic mode (north_star bar: log-mel L1 <= 1e-3).


def _mel_input(B, T, seed=0):
    rng = np.random.default_rng(seed)
    return (10.0 ** (rng.normal(size=(B, 1, T, 128)) * 1.2 - 2.5)).astype(np.float32)


@pytest.mark.parametrize("B,T", [(1, 101), (2, 64), (1, 130)])
def test_resunet_mel_vs_oracle(engine, unet_sd, B, T):
    from oracle import resunet
    mel = _mel_input(B, T)
    ref = resunet.generator_mel(unet_sd, torch.from_numpy(mel)).numpy()[:, 0]
    got = engine.resunet_mel(torch.from_numpy(mel[:, 0])).cpu().numpy()
    d = np.abs(got - ref)
    assert d.mean() < engine.tol['logmel_l1'], d.mean()
    assert d.max() < engine.tol['logmel_max'], d.max()
    # unet.py:78,99: the last mel bin of the UNet output is exactly 0 -> out = to_log(mel) there
    assert np.abs(got[..., 127] - np.log10(np.clip(mel[:, 0, :, 127], 1e-8, None))).max() < 1e-6
    assert engine.take_flags() == 0


def test_resunet_mel_same_bits_on_either_level_1_block_kernel(unet_sd):
    """The whole mel ResUNet with the persistent C = 32 block kernel (k_block2d32: 14 x 18 h tiles at 128 mel bins) and with
    k_resblock's 16 x 16 form (VFX_TUNE_OLD_BLOCK2D): same products summed in the same order per pixel, whatever the tiling --
    the outputs are equal bit for bit, on a batch with a dozen tiles per persistent block (40 clips of 333 frames)."""
    from voicefixer_main_amd import _lib
    from voicefixer_main_amd.engine import Engine, MODEL_UNET_MEL
    mel = torch.from_numpy(_mel_input(40, 333, seed=7)[:, 0])
    outs = []
    for tuning in (0, _lib.TUNE_OLD_BLOCK2D):
        eng = Engine("cuda:0", config={"precision": 1, "tuning": tuning})
        eng.load_state_dict(MODEL_UNET_MEL, unet_sd)
        outs.append(eng.resunet_mel(mel).cpu())
        assert eng.take_flags() == 0
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]), (outs[0] - outs[1]).abs().max().item()


def test_resunet_mel_same_bits_with_one_or_two_launches_per_upsampler(unet_sd):
    """The five 2 x upsamplers of the mel ResUNet (odd output widths) and of the spectrogram ResUNet (pruned, even widths) as one
    launch of four phases each (round 6) and as two launches of two phases (VFX_TUNE_TWO_LAUNCH_UPSAMPLERS): equal outputs bit for bit."""
    from voicefixer_main_amd import _lib, synth
    from voicefixer_main_amd.engine import Engine, MODEL_UNET_MEL, MODEL_UNET_SPEC
    mel = torch.from_numpy(_mel_input(5, 301, seed=9)[:, 0])
    wav = torch.from_numpy(synth.make_clips(2, 1.3, seed=12)[:, 0])
    outs = []
    for tuning in (0, _lib.TUNE_TWO_LAUNCH_UPSAMPLERS):
        eng = Engine("cuda:0", config={"precision": 1, "tuning": tuning})
        eng.load_state_dict(MODEL_UNET_MEL, unet_sd)
        eng.load_state_dict(MODEL_UNET_SPEC, synth.make_resunet_state_dict(2))
        sp = eng.stft(wav.cuda(), want_mel=False, want_sp=True)["sp"]
        outs.append((eng.resunet_mel(mel).cpu(), eng.resunet_spec(sp, wav.cuda()).cpu()))
        assert eng.take_flags() == 0
    for a, b in zip(*outs):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b), (a - b).abs().max().item()


def test_resunet_negative_input_flag(engine):
    mel = _mel_input(1, 64)
    mel[0, 0, 3, 5] = -1.0
    engine.resunet_mel(torch.from_numpy(mel[:, 0]))
    assert engine.take_flags() & 1
    assert engine.take_flags() == 0
    # bin 127 never reaches the network (unet.py:78) but to_log's assert covers the whole tensor (pytorch_util.py:158)
    mel = _mel_input(1, 64)
    mel[0, 0, 40, 127] = -1e-3
    engine.resunet_mel(torch.from_numpy(mel[:, 0]))
    assert engine.take_flags() & 1


@pytest.mark.parametrize("B,T", [(1, 21), (2, 10)])
def test_vocoder_vs_oracle(engine, voc_sd, B, T):
    from oracle import vocoder as voc
    mel = _mel_input(B, T, seed=3)
    ref = voc.vocoder(voc_sd, torch.from_numpy(mel)).numpy()[:, 0]
    got = engine.vocoder(torch.from_numpy(mel[:, 0])).cpu().numpy()
    assert got.shape == ref.shape == (B, (T + T % 2 + 4) * 441)
    assert np.abs(got - ref).max() < engine.tol['voc_max'], np.abs(got - ref).max()


ALT_TABLES = {
    # the recalled table is scales (7, 7, 3, 3), depth (8, 8, 8, 8), 1024 channels, condnet 5 x 512, dilations 3^i (synth.VocoderSpec)
    "scales3737_depth4683": dict(upsample_scales=(3, 7, 3, 7), resstack_depth=(4, 6, 8, 3)),
    "half_width_base2": dict(channels=512, cond_channels=256, cond_layers=3, upsample_scales=(7, 3, 7, 3), resstack_depth=(2, 3, 5, 8),
                             dilation_base=2),
    "three_stages_977": dict(channels=512, cond_channels=384, cond_layers=4, upsample_scales=(9, 7, 7), resstack_depth=(3, 5, 4)),
}


@pytest.mark.parametrize("name", sorted(ALT_TABLES))
def test_vocoder_alternate_tables(engine, name):
    """The vocoder's layer table is RECALLED (oracle/vocoder.py: parity unpinned), so the kernels must follow `vfx_config`, not be
    specialised to the guess: the same plan builder on three other tables -- another scale order and other depths per stage; half
    the width, a shorter condnet, dilations 2^i and a 32-channel tail; three stages with a stride of 9 -- against the oracle run
    with the same table, in every arithmetic mode."""
    from oracle import vocoder as voc
    from voicefixer_main_amd import synth
    from voicefixer_main_amd.engine import Engine, MODEL_VOCODER
    t = ALT_TABLES[name]
    spec = type("Spec", (synth.VocoderSpec,), dict(t))
    ocfg = voc.VocoderConfig(**{k: (list(v) if isinstance(v, tuple) else v) for k, v in t.items()})
    n = len(spec.upsample_scales)
    sd = synth.make_vocoder_state_dict(11, spec)
    cfg = {"precision": engine.precision, "voc_channels": spec.channels, "voc_cond_channels": spec.cond_channels,
           "voc_cond_layers": spec.cond_layers, "voc_n_stages": n, "voc_scales": list(spec.upsample_scales) + [0] * (8 - n),
           "voc_depth": list(spec.resstack_depth) + [0] * (8 - n), "voc_dilation_base": spec.dilation_base}
    eng = Engine("cuda:0", config=cfg)
    eng.load_state_dict(MODEL_VOCODER, sd)
    B, T = 2, 23
    mel = _mel_input(B, T, seed=5)
    ref = voc.vocoder(sd, torch.from_numpy(mel), ocfg).numpy()[:, 0]
    got = eng.vocoder(torch.from_numpy(mel[:, 0])).cpu().numpy()
    assert got.shape == ref.shape == (B, (T + T % 2 + 4) * 441)
    assert np.abs(got - ref).max() < engine.tol['voc_max'], (name, np.abs(got - ref).max())
    assert eng.take_flags() & 3 == 0
    eng.close()


def test_restore_gsr_vs_oracle(engine, unet_sd, voc_sd):
    from oracle import pipeline
    from voicefixer_main_amd import synth
    wav = synth.make_clips(2, 0.8)
    ref = pipeline.restore_gsr(unet_sd, voc_sd, wav)
    out, logmel = engine.restore_gsr(torch.from_numpy(wav[:, 0]), want_logmel=True)
    out, logmel = out.cpu().numpy(), logmel.cpu().numpy()
    d = np.abs(logmel - ref["logmel"][:, 0])
    assert d.mean() < engine.tol['logmel_l1'], d.mean()
    assert out.shape == wav[:, 0].shape
    err = out - ref["wav"][:, 0]
    sisdr = 10 * np.log10((ref["wav"] ** 2).sum() / ((err ** 2).sum() + 1e-20))
    assert sisdr > engine.tol['sisdr'], sisdr


def test_poisoned_arena_stays_finite(unet_sd, voc_sd, monkeypatch):
    """The workspace arena is filled with NaN patterns before every plan run (vfx_config.tuning & VFX_TUNE_DEBUG_POISON_ARENA): every stage must
    produce finite output, i.e. no kernel reads a workspace buffer before something wrote it."""
    from voicefixer_main_amd import synth
    from voicefixer_main_amd.engine import Engine, MODEL_UNET_MEL, MODEL_UNET_SPEC, MODEL_VOCODER
    eng = Engine("cuda:0", config={"precision": 1, "tuning": 256})   # re-poisoned before every plan run, whatever ran before
    eng.load_state_dict(MODEL_UNET_MEL, unet_sd)
    eng.load_state_dict(MODEL_VOCODER, voc_sd)
    eng.load_state_dict(MODEL_UNET_SPEC, synth.make_resunet_state_dict(2))
    wav = torch.from_numpy(synth.make_clips(2, 1.0, seed=41)[:, 0]).cuda()
    sp = eng.stft(wav, want_mel=False, want_sp=True)["sp"]
    for y in (eng.resunet_spec(sp, wav), eng.restore_gsr(wav), eng.vocoder(eng.stft(wav)["mel"])):
        assert bool(torch.isfinite(y).all())


def test_sub_batches_match_single_launch(engine, monkeypatch):
    """Batches whose activations would pass 4 GiB run as consecutive sub-batches (32-bit offsets in the
    convolution kernels); forcing that path with VFX_MAX_CLIPS must not change a single bit."""
    from voicefixer_main_amd import synth
    wav = torch.from_numpy(synth.make_clips(3, 0.7, seed=51)[:, 0]).cuda()
    whole = engine.restore_gsr(wav).clone()
    monkeypatch.setenv("VFX_MAX_CLIPS", "2")
    parts = engine.restore_gsr(wav)
    mel = engine.stft(wav)["mel"]
    voc_parts = engine.vocoder(mel)
    monkeypatch.delenv("VFX_MAX_CLIPS")
    assert torch.equal(parts, whole)
    assert torch.equal(voc_parts, engine.vocoder(mel))


def _sisdr(est, ref):
    """Scale-invariant SDR in dB (the waveform bar of BASELINE.json's north_star)."""
    est, ref = est.astype(np.float64).ravel(), ref.astype(np.float64).ravel()
    a = (est * ref).sum() / ((ref * ref).sum() + 1e-30)
    e = est - a * ref
    return 10 * np.log10(((a * ref) ** 2).sum() / ((e * e).sum() + 1e-30))


def test_fp16_vocoder_mode(unet_sd, voc_sd):
    """precision = 2 (16-bit vocoder, BASELINE.json config 2): the ResUNet keeps split-bf16 operands and the log-mel
    bar; the vocoder multiplies fp16 operands with one MFMA per product, so its waveform is held to an SI-SDR bar
    against the fp32 oracle (and the log-mel of the restored waveform to the 1e-3 bar) instead of the 1e-4 absolute bar
    of the split mode.  Fresh handles: the first call on a device must already be right.  The measured figures go to
    gpurun_out/."""
    import json
    import os
    from oracle import pipeline
    from oracle import vocoder as voc
    from voicefixer_main_amd import synth
    from voicefixer_main_amd.engine import Engine, MODEL_UNET_MEL, MODEL_VOCODER
    eng = Engine("cuda:0", config={"precision": 2})
    eng.load_state_dict(MODEL_UNET_MEL, unet_sd)
    eng.load_state_dict(MODEL_VOCODER, voc_sd)
    ref_eng = Engine("cuda:0", config={"precision": 1})
    ref_eng.load_state_dict(MODEL_UNET_MEL, unet_sd)
    ref_eng.load_state_dict(MODEL_VOCODER, voc_sd)
    res = {}
    # vocoder alone, oracle mel in
    mel = _mel_input(2, 40, seed=3)
    ref = voc.vocoder(voc_sd, torch.from_numpy(mel)).numpy()[:, 0]
    got = eng.vocoder(torch.from_numpy(mel[:, 0])).cpu().numpy()
    got1 = ref_eng.vocoder(torch.from_numpy(mel[:, 0])).cpu().numpy()
    res["vocoder_sisdr_db"] = _sisdr(got, ref)
    res["vocoder_max_abs"] = float(np.abs(got - ref).max())
    res["vocoder_ref_peak"] = float(np.abs(ref).max())
    res["vocoder_sisdr_db_split"] = _sisdr(got1, ref)
    # whole restore
    wav = synth.make_clips(2, 0.8)
    r = pipeline.restore_gsr(unet_sd, voc_sd, wav)
    out, logmel = eng.restore_gsr(torch.from_numpy(wav[:, 0]), want_logmel=True)
    out, logmel = out.cpu().numpy(), logmel.cpu().numpy()
    res["restore_logmel_l1"] = float(np.abs(logmel - r["logmel"][:, 0]).mean())
    res["restore_sisdr_db"] = _sisdr(out, r["wav"][:, 0])
    out1 = ref_eng.restore_gsr(torch.from_numpy(wav[:, 0])).cpu().numpy()
    res["restore_sisdr_db_split"] = _sisdr(out1, r["wav"][:, 0])
    # log-mel of the restored waveform against the log-mel of the oracle's waveform
    from oracle import dsp
    _, m_got = dsp.wav_to_mel(out.astype(np.float64)[:, None])
    _, m_ref = dsp.wav_to_mel(r["wav"].astype(np.float64))
    res["restore_out_logmel_l1"] = float(np.abs(np.log10(np.clip(m_got, 1e-8, None)) - np.log10(np.clip(m_ref, 1e-8, None))).mean())
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_prec2.json", "w") as f:
        json.dump({k: float(v) for k, v in res.items()}, f, indent=1)
    print(res)
    assert res["restore_logmel_l1"] < 2e-4, res          # the ResUNet is still split-bf16: bar 1e-3
    assert res["vocoder_sisdr_db"] > 52.0, res              # stated waveform bar of the 16-bit vocoder
    assert res["restore_sisdr_db"] > 52.0, res
    assert res["restore_out_logmel_l1"] < 1e-3, res         # log-mel of the restored waveform: the north-star bar
    assert res["vocoder_sisdr_db_split"] > 60.0, res


def _rescaled_vocoder(voc_sd, s):
    """The same function with an internal tensor s times larger: the k7 convolution in front of the first upsampler
    (weights and bias) times s, the upsampler's weights divided by s (LeakyReLU is positively homogeneous)."""
    sd = {k: v.clone() for k, v in voc_sd.items()}
    sd["generator.1.weight"] *= s
    sd["generator.1.bias"] *= s
    sd["generator.3.layer.weight"] /= s
    return sd


def test_fp16_vocoder_dynamic_range_flag_and_rerun(unet_sd, voc_sd):
    """fp16 operands have 5 exponent bits: the 16-bit vocoder (precision 2) must hold its waveform bar over a wide range
    of activation magnitudes, and must SAY so when an activation leaves the fp16 range instead of clamping silently:
    the kernels raise VFX_FLAG_F16_SATURATED and the model-level call re-runs on split-bf16 operands.  Also run: weights
    WITHOUT the x0.25 damping of the residual branches that the other tests' synthetic vocoder uses.  Figures go to
    gpurun_out/fp16_robustness.json."""
    import json
    import os
    import warnings
    from oracle import vocoder as voc
    from voicefixer_main_amd import _lib, models, synth
    from voicefixer_main_amd.engine import Engine, MODEL_UNET_MEL, MODEL_VOCODER
    mel = _mel_input(2, 40, seed=3)
    mel_t = torch.from_numpy(mel[:, 0])
    res = {}
    eng = Engine("cuda:0", config={"precision": 2})
    eng.load_state_dict(MODEL_UNET_MEL, unet_sd)
    ref0 = voc.vocoder(voc_sd, torch.from_numpy(mel)).numpy()[:, 0]
    for name, s in (("x1", 1.0), ("up60dB", 1e3), ("down60dB", 1e-3)):
        eng.load_state_dict(MODEL_VOCODER, _rescaled_vocoder(voc_sd, s))
        got = eng.vocoder(mel_t).cpu().numpy()
        flags = eng.take_flags()
        res["sisdr_db_" + name] = _sisdr(got, ref0)
        res["flags_" + name] = flags
        assert flags == 0, (name, flags)
    assert res["sisdr_db_x1"] > 50 and res["sisdr_db_up60dB"] > 50 and res["sisdr_db_down60dB"] > 50, res
    # 100 dB down the k7 convolution's weights sit at 3e-7: two bits in fp16 (17.7 dB).  No kernel sees a WEIGHT leave the
    # fp16 range, so the library checks them when they are packed: every call on such a weight set raises the flag
    eng.load_state_dict(MODEL_VOCODER, _rescaled_vocoder(voc_sd, 1e-5))
    got = eng.vocoder(mel_t).cpu().numpy()
    res["sisdr_db_down100dB"] = _sisdr(got, ref0)
    assert eng.take_flags() & _lib.FLAG_F16_SATURATED
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        out = models.VoiceFixer(None, channels=1, engine=eng).vocoder(torch.from_numpy(mel).cuda())[:, 0].cpu().numpy()
    res["sisdr_db_down100dB_rerun"] = _sisdr(out, ref0)
    assert res["sisdr_db_down100dB_rerun"] > 70, res
    # beyond the range: flagged, and the model-level call answers with the split-bf16 result
    big = _rescaled_vocoder(voc_sd, 3e5)
    eng.load_state_dict(MODEL_VOCODER, big)
    clamped = eng.vocoder(mel_t).cpu().numpy()
    flags = eng.take_flags()
    res["sisdr_db_up110dB_clamped"] = _sisdr(clamped, ref0)
    assert flags & _lib.FLAG_F16_SATURATED, flags
    m = models.VoiceFixer(None, channels=1, engine=eng)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = m.vocoder(torch.from_numpy(mel).cuda())[:, 0].cpu().numpy()
    assert any("fp16 range" in str(x.message) for x in w)
    res["sisdr_db_up110dB_rerun"] = _sisdr(out, ref0)
    assert res["sisdr_db_up110dB_rerun"] > 70, res
    assert eng.take_flags() == 0
    # undamped residual branches (every ResStack conv at full Kaiming-uniform gain): larger, growing residual stream
    gen = torch.Generator().manual_seed(5)
    und = {k: v.clone() for k, v in voc_sd.items()}
    for k in und:
        if ".res_layers." in k and k.endswith(".3.weight"):
            und[k] = und[k] * 4.0          # undo synth.make_vocoder_state_dict's gain = 0.25
    ref_u = voc.vocoder(und, torch.from_numpy(mel)).numpy()[:, 0]
    res["undamped_ref_peak"] = float(np.abs(ref_u).max())
    for prec in (2, 1):
        e = Engine("cuda:0", config={"precision": prec})
        e.load_state_dict(MODEL_VOCODER, und)
        got = e.vocoder(mel_t).cpu().numpy()
        res["undamped_sisdr_db_p%d" % prec] = _sisdr(got, ref_u)
        res["undamped_flags_p%d" % prec] = e.take_flags()
    assert res["undamped_flags_p2"] == 0 and res["undamped_sisdr_db_p2"] > 45 and res["undamped_sisdr_db_p1"] > 60, res
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/fp16_robustness.json", "w") as f:
        json.dump({k: float(v) for k, v in res.items()}, f, indent=1)
    print(res)


# vfx_config.tuning (include/vfx.h): every kernel-selection switch of the product path, exercised in process.  Each bit
# replaces one kernel family by an older / simpler form of the same arithmetic; the results must stay within the mode's bars.
TUNING = [("NO_FUSED_STACKS", 1, 2), ("NO_FUSED_WIDE", 2, 2), ("NO_FUSED_UNET", 4, 1), ("NO_PERSISTENT_C64", 8, 2),
          ("NO_PAIRS", 16, 2), ("NO_SPLITK", 32, 1), ("F32_TRUNK", 64, 2), ("SMALL_2D_TILES", 128, 1), ("NO_FUSED_STACKS", 1, 1),
          ("DEBUG_POISON_ARENA", 256, 2), ("NO_FUSED_UPSAMPLERS", 512, 2), ("OLD_BLOCK2D", 1024, 1),
          ("TWO_LAUNCH_UPSAMPLERS", 2048, 1)]


@pytest.mark.parametrize("name,bit,precision", TUNING, ids=["%s-p%d" % (n, p) for n, _, p in TUNING])
def test_tuning_switches(name, bit, precision, unet_sd, voc_sd, capfd):
    """One Engine per switch: the whole restore path (mel ResUNet + vocoder at 3 x 1.5 s: T' = 7350 at C = 256, 66 150 at
    C = 64 -- several tiles per block, folded dilations) and a single wide / C = 64 / C = 128 layer against the oracle, and
    vfx_create's announcement of the non-default selection."""
    from oracle import pipeline
    from voicefixer_main_amd import synth
    from voicefixer_main_amd.engine import Engine, MODEL_UNET_MEL, MODEL_VOCODER
    from conftest import TOL
    tol = TOL[precision]
    eng = Engine("cuda:0", config={"precision": precision, "tuning": bit})
    assert "VFX_TUNE_" + name in capfd.readouterr().err
    eng.load_state_dict(MODEL_UNET_MEL, unet_sd)
    eng.load_state_dict(MODEL_VOCODER, voc_sd)
    wav = synth.make_clips(3, 1.5, seed=77)
    ref = _tuning_oracle()
    out, logmel = eng.restore_gsr(torch.from_numpy(wav[:, 0]), want_logmel=True)
    d = np.abs(logmel.cpu().numpy() - ref["logmel"][:, 0])
    assert d.mean() < tol["logmel_l1"] and d.max() < tol["logmel_max"], (d.mean(), d.max())
    assert _sisdr(out.cpu().numpy(), ref["wav"][:, 0]) > tol["sisdr"]
    assert eng.take_flags() == 0
    # the same kernel selection on a batch of clips of UNEQUAL length (vfx_restore_gsr_varlen: every launch of the selected forms
    # takes the per-clip lengths): three lengths in one padded-frame bucket, each equal to its own batch-of-one call.  The one
    # combination the library refuses: the persistent C = 64 kernel on the fp32 trunk (no register left for the length).
    lens = [66150, 61000, 57000]
    x = torch.zeros((3, lens[0]))
    for j, L in enumerate(lens):
        x[j, :L] = torch.from_numpy(wav[j, 0, :L])
    if name == "F32_TRUNK":
        with pytest.raises(RuntimeError, match="fp16 trunk"):
            eng.restore_gsr_varlen(x, lens)
        eng.take_flags()
        return
    got = eng.restore_gsr_varlen(x, lens)
    for j, L in enumerate(lens):
        assert torch.equal(got[j, :L], eng.restore_gsr(x[j:j + 1, :L].contiguous())[0]), (name, j)
    assert eng.take_flags() & 3 == 0


_TUNING_REF = {}


def _tuning_oracle():
    if not _TUNING_REF:
        from oracle import pipeline
        from voicefixer_main_amd import synth
        _TUNING_REF.update(pipeline.restore_gsr(synth.make_resunet_state_dict(0), synth.make_vocoder_state_dict(1),
                                                synth.make_clips(3, 1.5, seed=77)))
    return _TUNING_REF
