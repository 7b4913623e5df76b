"""GPU parity tests of the individual HIP kernels against the CPU oracle (through the C ABI)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _nchw(y):
    return y.permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 9, 127, 32, 32), (1, 20, 7, 64, 128), (2, 5, 1, 384, 384),
                                            (1, 16, 63, 32, 64), (3, 4, 3, 96, 256),
                                            (1, 188, 3, 64, 96), (2, 94, 1, 96, 64)])      # tall narrow: fewer rows per tile (round 5)
def test_conv3x3_prologue_residual(engine, B, H, W, Cin, Cout):
    """3x3 conv with BN-affine + LeakyReLU prologue, bias and residual epilogue vs fp64 torch."""
    x = _rand((B, Cin, H, W), 1)
    w = _rand((Cout, Cin, 3, 3), 2, 0.1)
    scale = torch.rand(Cin, generator=torch.Generator().manual_seed(3)) + 0.5
    shift = _rand((Cin,), 4, 0.2)
    bias = _rand((Cout,), 5, 0.1)
    res = _rand((B, Cout, H, W), 6)
    a = F.leaky_relu(x.double() * scale.double()[None, :, None, None] + shift.double()[None, :, None, None], 0.01)
    ref = F.conv2d(a, w.double(), bias.double(), padding=1) + res.double()
    y = engine.op_conv(_nhwc(x), w.numpy(), scale.numpy(), shift.numpy(), act=1, slope=0.01, bias=bias.numpy(),
                       residual=_nhwc(res))
    err = (_nchw(y.cpu()).double() - ref).abs().max().item()
    assert err < engine.tol['conv'] * max(1.0, ref.abs().max().item()), err


def test_conv3x3_narrow_images_sweep(engine):
    """Seeded sweep of images NARROWER than their tile (1-15 columns, 1-200 rows): plan_conv clips the patch window to the image's own
    columns there (round 6: level 6 of the mel ResUNet, 32 x 3 pixels, in one tile), the tile's dead columns read neighbouring patch rows
    into accumulator columns nobody stores -- every shape against fp64 torch, with split-K where the planner chooses it (384 channels)."""
    rng = np.random.default_rng(606)
    shapes = [(int(rng.integers(1, 4)), int(rng.integers(1, 201)), int(rng.integers(1, 16)), int(rng.choice([32, 64, 384])), int(rng.choice([32, 64, 96])))
              for _ in range(16)] + [(16, 32, 3, 384, 384), (2, 34, 3, 64, 32), (1, 33, 5, 32, 32), (3, 30, 3, 96, 64)]
    for i, (B, H, W, Cin, Cout) in enumerate(shapes):
        x = _rand((B, Cin, H, W), 100 + i)
        w = _rand((Cout, Cin, 3, 3), 200 + i, 0.05)
        scale = torch.rand(Cin, generator=torch.Generator().manual_seed(300 + i)) + 0.5
        shift = _rand((Cin,), 400 + i, 0.2)
        a = F.leaky_relu(x.double() * scale.double()[None, :, None, None] + shift.double()[None, :, None, None], 0.01)
        ref = F.conv2d(a, w.double(), padding=1)
        y = engine.op_conv(_nhwc(x), w.numpy(), scale.numpy(), shift.numpy(), act=1, slope=0.01)
        assert torch.isfinite(y).all(), (B, H, W, Cin, Cout)
        err = (_nchw(y.cpu()).double() - ref).abs().max().item()
        assert err < engine.tol['conv'] * max(1.0, ref.abs().max().item()), ((B, H, W, Cin, Cout), err)


def test_conv1d_dilated_and_reflect(engine):
    B, T, C = 2, 300, 64
    x = _rand((B, C, T), 11)
    w3 = _rand((C, C, 3), 12, 0.1)
    bias = _rand((C,), 13, 0.1)
    for d in (1, 3, 27, 243):
        ref = F.conv1d(F.leaky_relu(x.double(), 0.01), w3.double(), bias.double(), padding=d, dilation=d)
        y = engine.op_conv(x.permute(0, 2, 1)[:, None].contiguous(), w3[:, :, None, :].numpy(), act=1, slope=0.01,
                           bias=bias.numpy(), dil_w=d)
        err = (y.cpu()[:, 0].permute(0, 2, 1).double() - ref).abs().max().item()
        assert err < engine.tol['conv'], (d, err)
    w7 = _rand((128, C, 7), 14, 0.1)
    ref = F.conv1d(F.pad(F.elu(x.double()), (3, 3), mode="reflect"), w7.double())
    y = engine.op_conv(x.permute(0, 2, 1)[:, None].contiguous(), w7[:, :, None, :].numpy(), act=2, reflect_w=True)
    err = (y.cpu()[:, 0].permute(0, 2, 1).double() - ref).abs().max().item()
    assert err < engine.tol['conv'] * max(1.0, ref.abs().max().item()), err


def _resblock_ref(x, w1, b1, w2, b2, d, slope):
    h = F.conv1d(F.leaky_relu(x.double(), slope), w1.double(), b1.double(), padding=d, dilation=d)
    return x.double() + F.conv1d(F.leaky_relu(h, slope), w2.double(), b2.double(), padding=1)


@pytest.mark.parametrize("C,T,fused", [(64, 1000, True), (128, 777, True), (64, 90, True), (96, 500, False), (256, 333, False),
                                       (256, 1100, True), (128, 90, True), (256, 90, True), (128, 3, True)])
def test_resblock_layer(engine, C, T, fused):
    """One ResStack layer (oracle/vocoder.py): fused k_resblock and the two-launch form with the activated
    intermediate tensor, over the vocoder's dilations (plain tiles up to 27, folded geometry beyond, also d > T; T = 90 / 3 with
    two clips: ONE tile per clip -- the tile -> clip split of the 4-wave kernels once divided by a reciprocal that overflowed there).
    C = 256 fused = the wide layer of the 16-bit mode on the two-form trunk (resblock_w64.hip; the entry point also checks its
    activated fp16 output against fp16(LeakyReLU(y))); its other forms: test_wide_layer_forms."""
    if fused and engine.tol['name'] == 'fp32':
        pytest.skip("the fused kernel has no fp32 form; fp32 plans use the two-launch form")
    if fused and C == 256 and engine.tol['name'] != 'fp16-vocoder':
        pytest.skip("the fused wide layer exists in the 16-bit mode only")
    B = 2
    x = _rand((B, C, T), 21)
    w1, w2 = _rand((C, C, 3), 22, 0.08), _rand((C, C, 3), 23, 0.08)
    b1, b2 = _rand((C,), 24, 0.1), _rand((C,), 25, 0.1)
    for d in (1, 3, 9, 27, 81, 243, 729, 2187):
        ref = _resblock_ref(x, w1, b1, w2, b2, d, 0.01)
        y = engine.op_resblock(x.permute(0, 2, 1).contiguous(), w1.numpy(), b1.numpy(), w2.numpy(), b2.numpy(), d, 0.01, fused)
        err = (y.cpu().permute(0, 2, 1).double() - ref).abs().max().item()
        assert err < engine.tol['conv'] * max(1.0, ref.abs().max().item()), (d, err)


@pytest.mark.parametrize("tuning,T", [(0, 1100), (0, 7350), (0, 50), (64, 1100), (64, 7350), (64, 50)],
                         ids=["f16-trunk", "f16-trunk-long", "f16-trunk-one-tile", "two-form", "two-form-long", "two-form-one-tile"])
def test_wide_layer_forms(tuning, T):
    """The two trunk forms of the C = 256 layer of the 16-bit mode (resblock_w64.hip; vfx_config.tuning: 0 = the activated fp16
    tensor is the whole trunk and the residual is LeakyReLU^-1 of it, VFX_TUNE_F32_TRUNK = raw fp32 x / y beside xa / ya) against
    the float64 layer, all eight dilations (1-D tiles up to 9, folded rows above, d > T), three clips of unequal tile phase; the
    op checks the activated fp16 output against fp16(LeakyReLU(y)) itself (two-form) or returns LeakyReLU^-1(ya) (fp16 trunk)."""
    from voicefixer_main_amd.engine import Engine
    from conftest import TOL
    eng = Engine("cuda:0", config={"precision": 2, "tuning": tuning})
    B, C = 3, 256
    x = _rand((B, C, T), 61)
    w1, w2 = _rand((C, C, 3), 62, 0.05), _rand((C, C, 3), 63, 0.05)
    b1, b2 = _rand((C,), 64, 0.1), _rand((C,), 65, 0.1)
    for d in (1, 3, 9, 27, 81, 243, 729, 2187):
        ref = _resblock_ref(x, w1, b1, w2, b2, d, 0.01)
        y = eng.op_resblock(x.permute(0, 2, 1).contiguous(), w1.numpy(), b1.numpy(), w2.numpy(), b2.numpy(), d, 0.01, True)
        err = (y.cpu().permute(0, 2, 1).double() - ref).abs().max().item()
        assert err < TOL[2]['conv'] * max(1.0, ref.abs().max().item()), (tuning, d, err)
    assert eng.take_flags() == 0


def test_resblock_layer_long_sequence(engine):
    """C = 64 over sequences long enough that a persistent block of the register-weights kernel (resblock_rw.hip, 16-bit
    mode) walks SEVERAL tiles -- prefetched patches, the kept residual and the LDS overlays are reused across tiles; clips of
    unequal tile phase (T not a multiple of the tile).  The other modes run k_resblock over the same shapes."""
    if engine.tol['name'] == 'fp32':
        pytest.skip("the fused kernel has no fp32 form")
    B, C, T = 3, 64, 70001
    x = _rand((B, C, T), 41)
    w1, w2 = _rand((C, C, 3), 42, 0.08), _rand((C, C, 3), 43, 0.08)
    b1, b2 = _rand((C,), 44, 0.1), _rand((C,), 45, 0.1)
    for d in (1, 27, 81, 2187):
        ref = _resblock_ref(x, w1, b1, w2, b2, d, 0.01)
        y = engine.op_resblock(x.permute(0, 2, 1).contiguous(), w1.numpy(), b1.numpy(), w2.numpy(), b2.numpy(), d, 0.01, True)
        err = (y.cpu().permute(0, 2, 1).double() - ref).abs().max().item()
        assert err < engine.tol['conv'] * max(1.0, ref.abs().max().item()), (d, err)


@pytest.mark.parametrize("C,T", [(64, 70001), (64, 300), (64, 3), (128, 70001), (128, 300), (128, 3)])
def test_resblock_layer_pair(engine, C, T):
    """Two consecutive ResStack layers as ONE launch (16-bit mode; C = 64: resblock_rw.hip, C = 128: resblock_r128.hip): the pairs
    the vocoder plan forms -- dilations (1, 3) and, at C = 64, (9, 27) -- and one more, over several tiles per block, a single
    partial tile and a sequence shorter than every halo.  The tensor between the layers exists only inside the kernel."""
    if engine.tol['name'] != 'fp16-vocoder':
        pytest.skip("layer pairs exist in the 16-bit mode only")
    B = 3
    x = _rand((B, C, T), 51)
    la = (_rand((C, C, 3), 52, 0.08), _rand((C,), 53, 0.1), _rand((C, C, 3), 54, 0.08), _rand((C,), 55, 0.1))
    lb = (_rand((C, C, 3), 56, 0.08), _rand((C,), 57, 0.1), _rand((C, C, 3), 58, 0.08), _rand((C,), 59, 0.1))
    for da, db in (((1, 3), (9, 27), (3, 9)) if C == 64 else ((1, 3), (2, 4), (16, 1))):
        y1 = _resblock_ref(x, la[0], la[1], la[2], la[3], da, 0.01)
        ref = _resblock_ref(y1, lb[0], lb[1], lb[2], lb[3], db, 0.01)
        y = engine.op_resblock_pair(x.permute(0, 2, 1).contiguous(), [a.numpy() for a in la], da, [a.numpy() for a in lb], db, 0.01)
        err = (y.cpu().permute(0, 2, 1).double() - ref).abs().max().item()
        assert err < 2 * engine.tol['conv'] * max(1.0, ref.abs().max().item()), (da, db, err)


@pytest.mark.parametrize("C", [64, 128])
def test_f32_trunk_layers(C):
    """VFX_TUNE_F32_TRUNK: the round-3 form of the 16-bit ResStack kernels (raw fp32 x in, raw fp32 y out) stays selectable;
    single layers (1-D and folded tiles) and the (1, 3) pair against the float64 layers.  The `engine` fixture's 16-bit mode
    runs the same tests on the fp16 trunk (the default)."""
    from voicefixer_main_amd.engine import Engine
    from conftest import TOL
    eng = Engine("cuda:0", config={"precision": 2, "tuning": 64})
    B, T = 3, 7001
    x = _rand((B, C, T), 71)
    la = (_rand((C, C, 3), 72, 0.08), _rand((C,), 73, 0.1), _rand((C, C, 3), 74, 0.08), _rand((C,), 75, 0.1))
    lb = (_rand((C, C, 3), 76, 0.08), _rand((C,), 77, 0.1), _rand((C, C, 3), 78, 0.08), _rand((C,), 79, 0.1))
    xc = x.permute(0, 2, 1).contiguous()
    for d in (1, 9, 81, 2187):
        ref = _resblock_ref(x, la[0], la[1], la[2], la[3], d, 0.01)
        y = eng.op_resblock(xc, la[0].numpy(), la[1].numpy(), la[2].numpy(), la[3].numpy(), d, 0.01, True)
        err = (y.cpu().permute(0, 2, 1).double() - ref).abs().max().item()
        assert err < TOL[2]['conv'] * max(1.0, ref.abs().max().item()), (C, d, err)
    y1 = _resblock_ref(x, la[0], la[1], la[2], la[3], 1, 0.01)
    ref = _resblock_ref(y1, lb[0], lb[1], lb[2], lb[3], 3, 0.01)
    y = eng.op_resblock_pair(xc, [a.numpy() for a in la], 1, [a.numpy() for a in lb], 3, 0.01)
    err = (y.cpu().permute(0, 2, 1).double() - ref).abs().max().item()
    assert err < 2 * TOL[2]['conv'] * max(1.0, ref.abs().max().item()), (C, err)
    assert eng.take_flags() == 0


@pytest.mark.parametrize("prune_w,H,W", [(False, 5, 1), (False, 10, 3), (True, 4, 16), (False, 6, 7), (False, 3, 30), (False, 17, 63),
                                        (True, 9, 5), (True, 2, 2)])
def test_conv_transpose2d(engine, prune_w, H, W):
    """The ResUNets' upsampler.  W >= 2 and H >= 2 run the product's form -- all FOUR parity classes as the phases of ONE launch
    (`TapConvParams::out_cmul` + `phase_rows`; at an odd width 2 W + 1, the mel ResUNet's, the odd column class is one column short),
    a block of k_conv covering up to four phases -- W = 1 the four parity launches."""
    B, Cin, Cout = 2, 64, 32
    x = _rand((B, Cin, H, W), 21)
    w = _rand((Cin, Cout, 3, 3), 22, 0.1)
    scale = torch.rand(Cin, generator=torch.Generator().manual_seed(23)) + 0.5
    shift = _rand((Cin,), 24, 0.2)
    a = F.relu(x.double() * scale.double()[None, :, None, None] + shift.double()[None, :, None, None])
    ref = F.conv_transpose2d(a, w.double(), stride=2)
    ref = ref[:, :, :-1, :-1] if prune_w else ref[:, :, :-1, :]
    y = engine.op_conv_transpose(_nhwc(x), w.numpy(), 2, prune_w=prune_w, scale=scale.numpy(), shift=shift.numpy(),
                                 act=1, slope=0.0)
    assert tuple(_nchw(y).shape) == tuple(ref.shape)
    err = (_nchw(y.cpu()).double() - ref).abs().max().item()
    assert err < engine.tol['conv'], err


@pytest.mark.parametrize("prune_w", [False, True])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 9, 6, 64, 32), (3, 40, 33, 64, 64), (1, 2, 2, 32, 32), (2, 63, 16, 128, 64), (5, 17, 129, 32, 32),
                                            (1, 12, 20, 64, 128), (2, 5, 9, 32, 96)])
def test_conv_transpose2d_one_launch_equals_two(engine, B, H, W, Cin, Cout, prune_w):
    """The upsampler as one launch of four phases whose blocks cover 1, 2 or 4 phases (round 6; 32 / 64 / 96 / 128 couts per phase)
    against the two launches of two phases each with one phase per block (VFX_TUNE_TWO_LAUNCH_UPSAMPLERS, rounds 3-4): the same stage
    tables and sums per output -- bit-identical, at an odd (2 W + 1) and at a pruned (2 W) output width."""
    from voicefixer_main_amd import _lib
    from voicefixer_main_amd.engine import Engine
    x = _nhwc(_rand((B, Cin, H, W), 25))
    w = _rand((Cin, Cout, 3, 3), 26, 0.1)
    scale = torch.rand(Cin, generator=torch.Generator().manual_seed(27)) + 0.5
    shift = _rand((Cin,), 28, 0.2)
    prec = {'fp32': 0, 'split-bf16': 1, 'fp16-vocoder': 2}[engine.tol['name']]
    two = Engine("cuda:0", config={"precision": prec, "tuning": _lib.TUNE_TWO_LAUNCH_UPSAMPLERS})
    kw = dict(prune_w=prune_w, scale=scale.numpy(), shift=shift.numpy(), act=1, slope=0.0)
    y1 = engine.op_conv_transpose(x, w.numpy(), 2, **kw)
    y2 = two.op_conv_transpose(x, w.numpy(), 2, **kw)
    assert tuple(y1.shape) == (B, 2 * H, 2 * W if prune_w else 2 * W + 1, Cout)
    assert torch.isfinite(y1).all()
    assert torch.equal(y1, y2), (y1 - y2).abs().max().item()


@pytest.mark.parametrize("s", [7, 3])
def test_conv_transpose1d(engine, s):
    B, Cin, Cout, T = 2, 64, 32, 50
    x = _rand((B, Cin, T), 31)
    w = _rand((Cin, Cout, 2 * s), 32, 0.1)
    bias = _rand((Cout,), 33, 0.1)
    ref = F.conv_transpose1d(F.leaky_relu(x.double(), 0.2), w.double(), bias.double(), stride=s,
                             padding=s // 2 + s % 2, output_padding=s % 2)
    y = engine.op_conv_transpose(x.permute(0, 2, 1)[:, None].contiguous(), w[:, :, None, :].numpy(), s, act=1,
                                 slope=0.2, bias=bias.numpy())
    got = y.cpu()[:, 0].permute(0, 2, 1).double()
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < engine.tol['conv']


def test_stft_mag_phase_mel(engine):
    from oracle import dsp
    from voicefixer_main_amd import synth
    wav = synth.make_clips(3, 0.75)[:, 0]        # (3, L), L = 33075 -> T = 76
    mag, cos, sin = dsp.spectrogram_phase(wav[:, None].astype(np.float64))
    mel = dsp.mel_project(mag, dsp.mel_filterbank().astype(np.float64))
    out = engine.stft(wav, want_mel=True, want_sp=True, want_phase=True)
    sp = out["sp"].cpu().numpy().astype(np.float64)
    scale = mag.max()
    assert np.abs(sp - mag[:, 0]).max() < 2e-6 * scale
    # phase compared through re/im so that near-zero bins do not dominate
    re_g, im_g = sp * out["cos"].cpu().numpy(), sp * out["sin"].cpu().numpy()
    assert np.abs(re_g - (mag * cos)[:, 0]).max() < 3e-6 * scale
    assert np.abs(im_g - (mag * sin)[:, 0]).max() < 3e-6 * scale
    melg = out["mel"].cpu().numpy().astype(np.float64)
    assert np.abs(melg - mel[:, 0]).max() < 1e-5 * mel.max()
    lg = engine.stft(wav, want_mel=True, log10_mel=True)["mel"].cpu().numpy()
    assert np.abs(lg - dsp.to_log(mel[:, 0])).max() < 1e-3
    assert np.abs(lg - dsp.to_log(mel[:, 0])).mean() < 1e-5


@pytest.mark.parametrize("tail", [0, 1, 200, 440])
def test_istft_roundtrip_and_oracle(engine, tail):
    """Whole output incl. the L mod 441 tail samples, against the oracle and torch.istft(length=L)
    (tools/dsp/base.py:196-200: `end = start + length`)."""
    from oracle import dsp
    from voicefixer_main_amd import synth
    L = 441 * 60 + tail
    wav = synth.make_clips(2, L / 44100.0)[:, 0]
    assert wav.shape[-1] == L
    re, im = dsp.stft(wav.astype(np.float64))
    got = engine.istft(re.astype(np.float32), im.astype(np.float32), L).cpu().numpy()
    ref = dsp.istft(re, im, L)
    assert np.abs(got - ref).max() < 5e-6
    ti = torch.istft(torch.complex(torch.from_numpy(re), torch.from_numpy(im)).transpose(1, 2), 2048, 441, 2048,
                     torch.hann_window(2048, periodic=True, dtype=torch.float64), center=True, length=L).numpy()
    assert np.abs(got - ti).max() < 5e-6
    if tail:
        assert np.abs(got[:, -tail:] - ti[:, -tail:]).max() < 5e-6 and np.abs(got[:, -tail:]).max() > 1e-3
    # STFT -> ISTFT perfect reconstruction over the WHOLE clip (tools/dsp/base.py:214-232)
    o = engine.stft(wav, want_mel=False, want_sp=True, want_phase=True)
    back = engine.istft(o["sp"] * o["cos"], o["sp"] * o["sin"], L).cpu().numpy()
    assert np.abs(back - wav).max() < 1e-5
    # T frames that do not reach L: zeros only past the end of the overlap-add buffer
    short = engine.istft(re[:, :20].astype(np.float32), im[:, :20].astype(np.float32), L).cpu().numpy()
    sref = dsp.istft(re[:, :20], im[:, :20], L)
    well = 441 * 19 + 1024 - 300      # up to 300 samples before the end of the last window, w^2 >= 0.04; past that the
    assert np.abs(short[:, :well] - sref[:, :well]).max() < 5e-6      # division by the vanishing envelope amplifies fp32 rounding
    assert np.abs(short[:, well:] - sref[:, well:]).max() < 0.05 * np.abs(sref).max()
    assert np.all(short[:, 1024 + 441 * 19:] == 0)


def test_spectrogram_phase_eps(engine):
    """FDomainHelper.spectrogram_phase(input, eps) (fDomainHelper.py:60-65): the clamp is on the POWER and is the
    caller's; eps = 0 (that method's default) leaves exactly silent bins at magnitude 0 with NaN phases."""
    from oracle import dsp
    from voicefixer_main_amd import synth
    from voicefixer_main_amd.models import FDomainHelper
    fh = FDomainHelper(engine)
    wav = synth.make_clips(2, 0.4)[:, 0]
    wav[1] = 0.0                                               # a silent clip: every bin exactly zero
    x = torch.from_numpy(wav).cuda()
    for eps in (1e-8, 1e-4, 0.0):
        sp, cos, sin = fh.spectrogram_phase(x, eps=eps)
        mag, c, s_ = dsp.spectrogram_phase(wav[:, None].astype(np.float64), eps=max(eps, 1e-300))
        assert sp.shape == (2, 1, wav.shape[-1] // 441 + 1, 1025)
        assert np.abs(sp.cpu().numpy()[0] - mag[0]).max() < 3e-6 * mag.max()
        if eps > 0:
            assert float(sp.min()) >= np.sqrt(eps) * (1 - 1e-6)
            assert torch.isfinite(cos).all() and torch.isfinite(sin).all()
            assert float(sp[1].max()) <= np.sqrt(eps) * (1 + 1e-6)
        else:
            assert float(sp[1].abs().max()) == 0.0 and torch.isnan(cos[1]).all() and torch.isnan(sin[1]).all()
            assert torch.isfinite(cos[0]).all()
    sp8, _, _ = fh.wav_to_spectrogram_phase(x[:, None])        # the handlers' call: eps = 1e-8
    assert float(sp8.min()) >= 1e-4 * (1 - 1e-6)
    with pytest.raises(ValueError):
        fh.spectrogram_phase(x, eps=-1.0)


def test_mel_project(engine):
    from oracle import dsp
    sp = np.abs(np.random.default_rng(0).normal(size=(2, 1, 7, 1025))).astype(np.float32)
    ref = dsp.mel_project(sp.astype(np.float64), dsp.mel_filterbank().astype(np.float64))
    got = engine.mel_project(torch.from_numpy(sp)).cpu().numpy()
    assert np.abs(got - ref).max() < 1e-5 * ref.max()


@pytest.mark.parametrize("B,H,W", [(2, 40, 127), (3, 7, 5), (1, 100, 128), (5, 14, 14), (2, 61, 30), (1, 16, 16), (19, 33, 45), (16, 310, 128), (16, 310, 126), (4, 24, 32)])
def test_block2d32_equals_the_one_tile_per_block_kernel(engine, B, H, W):
    """The persistent C = 32 block kernel (block2d32.hip: padded LDS rows, tables computed once per block, masks on border tiles
    only) against k_resblock's 16 x 16 form of the same block (VFX_TUNE_OLD_BLOCK2D): the same products summed in the same order,
    so every output must be bit-identical -- interior tiles, all four borders, images smaller than a tile, fewer tiles than
    blocks and (16 x 310 x 128 on 14 x 18 h tiles, 16 x 310 x 126 on 16 x 16 ones: the planner picks the grid with fewer tiles)
    several tiles per block, where the loop-carried state and the hazards between a tile's last MFMAs and its staged stores show."""
    if engine.tol['name'] == 'fp32':
        pytest.skip("the fused block exists for the split-bf16 ResUNet arithmetic (precision 1 and 2) only")
    from voicefixer_main_amd import _lib
    from voicefixer_main_amd.engine import Engine
    C = 32
    x = _rand((B, C, H, W), 51) * 3.0
    w1, w2 = _rand((C, C, 3, 3), 52, 0.06), _rand((C, C, 3, 3), 53, 0.06)
    g = torch.Generator().manual_seed(54)
    sc1, sc2 = torch.rand(C, generator=g) + 0.5, torch.rand(C, generator=g) + 0.5
    sh1, sh2 = _rand((C,), 55, 0.2), _rand((C,), 56, 0.2)
    args = (w1.numpy(), sc1.numpy(), sh1.numpy(), w2.numpy(), sc2.numpy(), sh2.numpy(), 0.01)
    y = engine.op_block2d(_nhwc(x), *args).cpu()
    old = Engine("cuda:0", config={"precision": 1, "tuning": _lib.TUNE_OLD_BLOCK2D})
    y_old = old.op_block2d(_nhwc(x), *args).cpu()
    assert torch.isfinite(y).all()
    assert torch.equal(y, y_old), (y - y_old).abs().max().item()
    again = engine.op_block2d(_nhwc(x), *args).cpu()  # no state between calls
    assert torch.equal(y, again)


def test_block2d32_random_shapes_equal_the_one_tile_per_block_kernel(engine):
    """Seeded sweep of image shapes (1-20 images, 1-400 rows, 1-200 columns, plus the benched 16 x 1001 x 128) through both C = 32 block
    kernels: bit-identical outputs whatever the tile grid the planner picks, the number of tiles per persistent block and the overhang at
    the right and bottom borders."""
    if engine.tol['name'] == 'fp32':
        pytest.skip("the fused block exists for the split-bf16 ResUNet arithmetic (precision 1 and 2) only")
    from voicefixer_main_amd import _lib
    from voicefixer_main_amd.engine import Engine
    C = 32
    rng = np.random.default_rng(2026)
    shapes = [(int(rng.integers(1, 21)), int(rng.integers(1, 401)), int(rng.integers(1, 201))) for _ in range(14)] + [(16, 1001, 128)]
    old = Engine("cuda:0", config={"precision": 1, "tuning": _lib.TUNE_OLD_BLOCK2D})
    w1, w2 = _rand((C, C, 3, 3), 62, 0.06), _rand((C, C, 3, 3), 63, 0.06)
    g = torch.Generator().manual_seed(64)
    sc1, sc2 = torch.rand(C, generator=g) + 0.5, torch.rand(C, generator=g) + 0.5
    sh1, sh2 = _rand((C,), 65, 0.2), _rand((C,), 66, 0.2)
    args = (w1.numpy(), sc1.numpy(), sh1.numpy(), w2.numpy(), sc2.numpy(), sh2.numpy(), 0.01)
    for i, (B, H, W) in enumerate(shapes):
        x = _nhwc(_rand((B, C, H, W), 70 + i) * 2.0)
        y = engine.op_block2d(x, *args)
        y_old = old.op_block2d(x, *args)
        assert torch.isfinite(y).all(), (B, H, W)
        assert torch.equal(y, y_old), ((B, H, W), (y - y_old).abs().max().item())


@pytest.mark.parametrize("C,H,W", [(32, 40, 127), (64, 33, 63), (32, 7, 5), (64, 130, 20)])
def test_fused_conv_block_res(engine, C, H, W):
    """One ConvBlockRes with identity shortcut as ONE launch (k_resblock, 2-D mode): bn1 -> lrelu -> 3x3 -> bn2 -> lrelu ->
    3x3 -> + x, against the same block in float64 torch; tiles that overhang every image border."""
    if engine.tol['name'] == 'fp32':
        pytest.skip("the fused block exists for the split-bf16 ResUNet arithmetic (precision 1 and 2) only")
    B = 2
    x = _rand((B, C, H, W), 41)
    w1, w2 = _rand((C, C, 3, 3), 42, 0.06), _rand((C, C, 3, 3), 43, 0.06)
    g = torch.Generator().manual_seed(44)
    sc1, sc2 = torch.rand(C, generator=g) + 0.5, torch.rand(C, generator=g) + 0.5
    sh1, sh2 = _rand((C,), 45, 0.2), _rand((C,), 46, 0.2)
    aff = lambda t, sc, sh: t * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]
    h = F.conv2d(F.leaky_relu(aff(x.double(), sc1, sh1), 0.01), w1.double(), padding=1)
    ref = F.conv2d(F.leaky_relu(aff(h, sc2, sh2), 0.01), w2.double(), padding=1) + x.double()
    y = engine.op_block2d(_nhwc(x), w1.numpy(), sc1.numpy(), sh1.numpy(), w2.numpy(), sc2.numpy(), sh2.numpy(), 0.01)
    err = (_nchw(y.cpu()).double() - ref).abs().max().item()
    assert err < engine.tol['conv'] * max(1.0, ref.abs().max().item()), err
