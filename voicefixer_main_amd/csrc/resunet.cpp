// resunet.cpp -- weights and launch plans of the two ResUNets.
//
//   mel ResUNet          models/components/unet.py:12-103     (B,1,T,128) log-mel -> residual
//   spectrogram ResUNet  models/components/unet_v2.py:21-148  (B,1,T,1025) magnitude -> magnitude
//
// Both share the trunk: 6 x EncoderBlockRes4B (modules.py:167-184), ConvBlockRes bottleneck,
// 6 x DecoderBlockRes4B (modules.py:186-220), after_conv_block1, after_conv2.  Activations live
// in the workspace arena as channels-last fp32 (B, H=T, W=F, C).  Every 3x3 / 1x1 / transposed
// convolution with Cin >= 32 is one launch (4 for a transposed conv) of the tap-convolution
// kernel; BatchNorm (eval mode) is folded into a per-channel affine applied, together with the
// LeakyReLU, while the input tile is staged into LDS; the residual add / 1x1 shortcut / channel
// concat never materialise.
#include <cmath>

#include "vfx_internal.h"

namespace vfx {

static const int kEncC[6] = {32, 64, 128, 256, 384, 384};                  // unet.py:22-33
static const int kDecIn[6] = {384, 384, 384, 256, 128, 64};                // unet.py:36-47
static const int kDecOut[6] = {384, 384, 256, 128, 64, 32};
static const float kBnEps = 1e-5f;
static const float kSlope = 0.01f;  // modules.py:265-266

namespace {

struct Staged {
  vfx_handle* h;
  int model;
  const HostTensor& get(const std::string& name) const {
    auto it = h->staged[model].find(name);
    VFX_CHECK(it != h->staged[model].end(), "missing tensor '%s' for model %d", name.c_str(), model);
    return it->second;
  }
  bool has(const std::string& name) const { return h->staged[model].count(name) != 0; }
};

std::vector<std::pair<int, int>> taps3x3() {
  std::vector<std::pair<int, int>> t;
  for (int kh = 0; kh < 3; ++kh)
    for (int kw = 0; kw < 3; ++kw) t.push_back({kh, kw});
  return t;
}

// eval-mode BatchNorm2d -> (scale, shift)
void fold_bn(const Staged& st, const std::string& p, int n, std::vector<float>& scale, std::vector<float>& shift) {
  const HostTensor &g = st.get(p + ".weight"), &b = st.get(p + ".bias"), &m = st.get(p + ".running_mean"),
                   &v = st.get(p + ".running_var");
  VFX_CHECK((int)g.data.size() == n && (int)v.data.size() == n, "%s: expected %d features", p.c_str(), n);
  scale.resize(n);
  shift.resize(n);
  for (int i = 0; i < n; ++i) {
    const float s = g.data[i] / std::sqrt(v.data[i] + kBnEps);
    scale[i] = s;
    shift[i] = b.data[i] - m.data[i] * s;
  }
}

void check_shape(const HostTensor& t, std::initializer_list<int64_t> shp, const std::string& name) {
  VFX_CHECK(t.shape == std::vector<int64_t>(shp), "tensor '%s' has an unexpected shape", name.c_str());
}

ConvBlockW load_block(const Staged& st, DeviceBlob& blob, const std::string& p, int cin, int cout, int nsrc) {
  const bool split = st.h->cfg.precision != 0;
  ConvBlockW w;
  w.cin = cin;
  w.cout = cout;
  w.nsrc = nsrc;
  std::vector<float> sc, sh;
  fold_bn(st, p + ".bn1", cin, sc, sh);
  w.bn1_scale = blob.upload(sc);
  w.bn1_shift = blob.upload(sh);
  fold_bn(st, p + ".bn2", cout, sc, sh);
  w.bn2_scale = blob.upload(sc);
  w.bn2_shift = blob.upload(sh);
  const HostTensor& c1 = st.get(p + ".conv1.weight");
  check_shape(c1, {cout, cin, 3, 3}, p + ".conv1.weight");
  const HostTensor& c2 = st.get(p + ".conv2.weight");
  check_shape(c2, {cout, cout, 3, 3}, p + ".conv2.weight");
  const int cs = cin / nsrc;
  if (cin >= kKC)
    for (int s = 0; s < nsrc; ++s) w.w1[s] = blob.upload(pack_conv(c1.data.data(), cout, cin, 3, 3, s * cs, cs, taps3x3(), split));
  w.w2 = blob.upload(pack_conv(c2.data.data(), cout, cout, 3, 3, 0, cout, taps3x3(), split));
  w.shortcut = st.has(p + ".shortcut.weight");
  VFX_CHECK(w.shortcut == (cin != cout), "%s: shortcut presence does not match channel counts", p.c_str());
  if (w.shortcut) {
    const HostTensor& ws = st.get(p + ".shortcut.weight");
    check_shape(ws, {cout, cin, 1, 1}, p + ".shortcut.weight");
    if (cin >= kKC)
      for (int s = 0; s < nsrc; ++s)
        w.wsc[s] = blob.upload(pack_conv(ws.data.data(), cout, cin, 1, 1, s * cs, cs, {{0, 0}}, split));
    w.bsc = blob.upload(st.get(p + ".shortcut.bias").data);
  }
  return w;
}

}  // namespace

std::shared_ptr<UNetWeights> build_unet_weights(vfx_handle* h, int model) {
  Staged st{h, model};
  auto W = std::make_shared<UNetWeights>();
  DeviceBlob& blob = h->blob;
  int cin = 1;
  for (int l = 0; l < 6; ++l) {
    const int c = kEncC[l];
    char p[64];
    for (int j = 0; j < 4; ++j) {
      snprintf(p, sizeof(p), "encoder_block%d.conv_block%d", l + 1, j + 1);
      W->enc[l][j] = load_block(st, blob, p, j == 0 ? cin : c, c, 1);
    }
    cin = c;
  }
  // Cin = 1 entry convs of encoder_block1.conv_block1
  {
    const std::string p = "encoder_block1.conv_block1";
    std::vector<float> sc, sh;
    fold_bn(st, p + ".bn1", 1, sc, sh);
    W->c1_scale = sc[0];
    W->c1_shift = sh[0];
    const HostTensor& c1 = st.get(p + ".conv1.weight");  // (32,1,3,3) -> [tap][32]
    std::vector<float> w(9 * 32);
    for (int n = 0; n < 32; ++n)
      for (int t = 0; t < 9; ++t) w[t * 32 + n] = c1.data[n * 9 + t];
    W->c1_w = blob.upload(w);
    W->c1_wsc = blob.upload(st.get(p + ".shortcut.weight").data);  // (32,1,1,1)
    W->c1_bsc = blob.upload(st.get(p + ".shortcut.bias").data);
  }
  W->bott = load_block(st, blob, "conv_block7", 384, 384, 1);
  for (int d = 0; d < 6; ++d) {
    DecoderW& D = W->dec[d];
    D.cin = kDecIn[d];
    D.cout = kDecOut[d];
    char p[64];
    snprintf(p, sizeof(p), "decoder_block%d", d + 1);
    std::vector<float> sc, sh;
    fold_bn(st, std::string(p) + ".bn1", D.cin, sc, sh);
    D.bn_scale = blob.upload(sc);
    D.bn_shift = blob.upload(sh);
    const HostTensor& wt = st.get(std::string(p) + ".conv1.weight");
    check_shape(wt, {D.cin, D.cout, 3, 3}, std::string(p) + ".conv1.weight");
    // output parity class (a, b) = (oh & 1, ow & 1): kernel rows kh == a (mod 2), cols kw == b (mod 2)
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b) {
        std::vector<std::pair<int, int>> taps;
        for (int kh = a; kh < 3; kh += 2)
          for (int kw = b; kw < 3; kw += 2) taps.push_back({kh, kw});
        D.wT[a * 2 + b] = blob.upload(pack_conv_transposed(wt.data.data(), D.cin, D.cout, 3, 3, taps, h->cfg.precision != 0));
      }
    for (int j = 0; j < 4; ++j) {
      char q[96];
      snprintf(q, sizeof(q), "%s.conv_block%d", p, j + 2);
      D.blocks[j] = load_block(st, blob, q, j == 0 ? 2 * D.cout : D.cout, D.cout, j == 0 ? 2 : 1);
    }
  }
  W->after = load_block(st, blob, "after_conv_block1", 32, 32, 1);
  const HostTensor& fw = st.get("after_conv2.weight");
  check_shape(fw, {1, 32, 1, 1}, "after_conv2.weight");
  W->final_w = blob.upload(fw.data);
  W->final_b = st.get("after_conv2.bias").data[0];
  return W;
}

// ---------------------------------------------------------------------------------------------
// plan construction
// ---------------------------------------------------------------------------------------------
namespace {

struct Act4 {  // an activation tensor in the arena
  size_t off = 0;
  int H = 0, W = 0, C = 0;
};

struct TrunkBuilder {
  PlanBuilder& pb;
  const UNetWeights& Wt;
  int B;

  Act4 make(int H, int W, int C) {
    Act4 a;
    a.H = H;
    a.W = W;
    a.C = C;
    a.off = pb.alloc_f((int64_t)B * H * W * C);
    return a;
  }

  static void set_taps3x3(TapSeg& s) {
    s.ntaps = 9;
    for (int kh = 0; kh < 3; ++kh)
      for (int kw = 0; kw < 3; ++kw) {
        s.dh[kh * 3 + kw] = kh - 1;
        s.dw[kh * 3 + kw] = kw - 1;
      }
  }

  TapConvParams base(const Act4& geo, int Cout, size_t out_off) {
    TapConvParams p{};
    p.B = B;
    p.Hi = p.Hg = p.Ho = geo.H;
    p.Wi = p.Wg = p.Wo = geo.W;
    p.Cout = Cout;
    p.sh = p.sw = 1;
    p.act_slope = 1.f;
    p.out = const_cast<float*>(rel_ptr(out_off));
    return p;
  }

  // y = ConvBlockRes(x) where x = cat(srcs) (1 or 2 sources); frees nothing.
  // `pre_h` / `pre_sc`: first block of the network, conv1 and shortcut already computed (Cin = 1).
  Act4 conv_block(const ConvBlockW& w, const Act4* srcs, int nsrc, const Act4* pre_h = nullptr,
                  const Act4* pre_sc = nullptr) {
    const Act4& g = srcs ? srcs[0] : *pre_h;
    // Identity-shortcut blocks of the C = 32 / 64 levels run as ONE launch of k_resblock's 2-D mode: h stays in LDS, 402
    // instead of 872 HBM bytes per output pixel (VFX_TUNE_NO_FUSED_UNET: the two-launch form).
    const bool fuse2d = !(pb.h->cfg.tuning & VFX_TUNE_NO_FUSED_UNET);
    if (fuse2d && srcs && nsrc == 1 && !w.shortcut && !pre_h && block2d_supported(w.cout) && pb.h->cfg.precision != 0) {
      Act4 y = make(g.H, g.W, w.cout);
      ResBlockParams rp{};
      rp.geo2d = 1;
      rp.x = rel_ptr(srcs[0].off);
      rp.y = const_cast<float*>(rel_ptr(y.off));
      rp.w1 = w.w1[0];
      rp.w2 = w.w2;
      rp.sc1 = w.bn1_scale;
      rp.sh1 = w.bn1_shift;
      rp.sc2 = w.bn2_scale;
      rp.sh2 = w.bn2_shift;
      rp.slope = kSlope;
      rp.B = B;
      rp.H = g.H;
      rp.W = g.W;
      rp.C = w.cout;
      pb.add_resblock(rp);
      return y;
    }
    // The two-source block with a 1x1 shortcut at C = 32 (decoder_block6.conv_block2): one launch too (resblock.hip, SC2)
#ifndef VFX_ABL_NO_SC2
    if (fuse2d && srcs && nsrc == 2 && w.shortcut && w.cout == 32 && srcs[0].C == 32 && srcs[1].C == 32 &&
        pb.h->cfg.precision != 0 && !(pb.h->cfg.tuning & VFX_TUNE_SMALL_2D_TILES)) {
      Act4 y = make(g.H, g.W, w.cout);
      ResBlockParams rp{};
      rp.geo2d = 1;
      rp.two_src = 1;
      rp.x = rel_ptr(srcs[0].off);
      rp.x2 = rel_ptr(srcs[1].off);
      rp.y = const_cast<float*>(rel_ptr(y.off));
      rp.w1 = w.w1[0];
      rp.w1x2 = w.w1[1];
      rp.w2 = w.w2;
      rp.wsc = w.wsc[0];
      rp.wsc2 = w.wsc[1];
      rp.bsc = w.bsc;
      rp.sc1 = w.bn1_scale;
      rp.sh1 = w.bn1_shift;
      rp.sc2 = w.bn2_scale;
      rp.sh2 = w.bn2_shift;
      rp.slope = kSlope;
      rp.B = B;
      rp.H = g.H;
      rp.W = g.W;
      rp.C = w.cout;
      pb.add_resblock(rp);
      return y;
    }
#endif
    Act4 hbuf;
    if (pre_h) {
      hbuf = *pre_h;
    } else {
      hbuf = make(g.H, g.W, w.cout);
      TapConvParams p = base(g, w.cout, hbuf.off);
      p.nseg = nsrc;
      int coff = 0;
      for (int s = 0; s < nsrc; ++s) {
        TapSeg& S = p.seg[s];
        S.src = rel_ptr(srcs[s].off);
        S.C = srcs[s].C;
        S.scale = w.bn1_scale + coff;
        S.shift = w.bn1_shift + coff;
        S.act = ACT_LEAKY;
        S.slope = kSlope;
        S.wt = w.w1[s];
        set_taps3x3(S);
        coff += srcs[s].C;
      }
      // h feeds conv2 only: store it ACTIVATED (bn2 affine + LeakyReLU applied once per element, operand form)
      p.out_act = p.out;
      p.out = nullptr;
      p.act_scale = w.bn2_scale;
      p.act_shift = w.bn2_shift;
      p.act_slope = kSlope;
      pb.add_conv(p);
    }
    Act4 y = make(g.H, g.W, w.cout);
    TapConvParams p = base(g, w.cout, y.off);
    TapSeg& S0 = p.seg[0];
    S0.src = rel_ptr(hbuf.off);
    S0.C = w.cout;
    if (pre_h) {  // h of the Cin = 1 entry block comes raw from k_conv_c1
      S0.scale = w.bn2_scale;
      S0.shift = w.bn2_shift;
      S0.act = ACT_LEAKY;
      S0.slope = kSlope;
    } else {
      S0.src_act = 1;
      S0.act = ACT_NONE;
      S0.slope = 1.f;
    }
    S0.wt = w.w2;
    set_taps3x3(S0);
    p.nseg = 1;
    if (pre_sc) {
      p.residual = rel_ptr(pre_sc->off);  // shortcut(x) incl. bias, computed by the Cin=1 kernel
    } else if (w.shortcut) {
      for (int s = 0; s < nsrc; ++s) {
        TapSeg& S = p.seg[p.nseg++];
        S.src = rel_ptr(srcs[s].off);
        S.C = srcs[s].C;
        S.scale = nullptr;
        S.shift = nullptr;
        S.act = ACT_NONE;
        S.slope = 1.f;
        S.wt = w.wsc[s];
        S.ntaps = 1;
        S.dh[0] = 0;
        S.dw[0] = 0;
      }
      p.bias = w.bsc;
    } else {
      p.residual = rel_ptr(srcs[0].off);
    }
    pb.add_conv(p);
    if (!pre_h) pb.free(hbuf.off);
    return y;
  }

  Act4 pool(const Act4& x) {
    Act4 y = make(x.H / 2, x.W / 2, x.C);
    Plan* pl = pb.plan;
    const int Bc = B;
    const Act4 xi = x;
    const size_t yo = y.off;
    pl->ops.push_back([=](const RunCtx& c) {
      launch_avgpool2(reinterpret_cast<const float*>(pl->bound_base + xi.off), Bc, xi.H, xi.W, xi.C,
                      reinterpret_cast<float*>(pl->bound_base + yo), c.stream);
    });
    return y;
  }

  // BN -> ReLU -> ConvTranspose2d(k3, s2, p0) -> prune (modules.py:205-214)
  Act4 upsample(const DecoderW& D, const Act4& x, bool prune_w) {
    Act4 y = make(2 * x.H, prune_w ? 2 * x.W : 2 * x.W + 1, D.cout);
    auto parity_taps = [](TapSeg& S, int a, int b) {
      S.ntaps = 0;
      for (int kh = a; kh < 3; kh += 2)
        for (int kw = b; kw < 3; kw += 2) {
          S.dh[S.ntaps] = -(kh / 2);  // oh = 2*ih + kh  ->  ih = i - kh/2 for oh = 2i + a
          S.dw[S.ntaps] = -(kw / 2);
          ++S.ntaps;
        }
    };
    // Round 6: ONE launch of four phases r = 2 a + b for either output width: the output is addressed in units of C channels at the TRUE
    // pixel index (TapConvParams::out_cmul: phase r lands at column 2 j + (r & 1)), phases 2 and 3 write the odd output rows
    // (TapConvParams::phase_rows), and a block of k_conv covers up to four phases (BN = 128 for 32 couts per phase): the patch is
    // staged once for all of them and x is read from HBM once.  VFX_TUNE_TWO_LAUNCH_UPSAMPLERS keeps the forms of rounds 3-4 below (one
    // launch per output row class, one phase per block); same stage tables, same sums: bit-identical.
    if (x.H >= 2 && x.W >= 2 && !(pb.h->cfg.tuning & (VFX_TUNE_NO_FUSED_UNET | VFX_TUNE_TWO_LAUNCH_UPSAMPLERS))) {
      TapConvParams p{};
      p.B = B;
      p.Hi = x.H;
      p.Wi = x.W;
      p.Ho = y.H;
      p.Wo = y.W;
      p.Cout = 4 * D.cout;
      p.out_cmul = D.cout;
      p.phase_rows = 1;
      p.sh = 2;
      p.sw = 2;
      p.oh0 = 0;
      p.ow0 = 0;
      p.Hg = (y.H + 1) / 2;            // (row class 1 has y.H / 2 rows: masked, oh < Ho)
      p.Wg = (y.W + 1) / 2;            // column class 0: W + 1 columns of an odd width 2 W + 1, W of a pruned one; class 1 masked (ow + 1 < Wo)
      p.out = const_cast<float*>(rel_ptr(y.off));
      p.act_slope = 1.f;
      p.nseg = 1;
      std::vector<TapSeg> phases(4);
      for (int r = 0; r < 4; ++r) {
        TapSeg& S = phases[r];
        S = TapSeg{};
        S.src = rel_ptr(x.off);
        S.C = x.C;
        S.scale = D.bn_scale;
        S.shift = D.bn_shift;
        S.act = ACT_LEAKY;
        S.slope = 0.f;  // ReLU
        S.wt = D.wT[r];
        parity_taps(S, r >> 1, r & 1);
      }
      p.seg[0] = phases[0];  // phase (0, 0) reads kh, kw in {0, 2}: the union of all four windows
      pb.add_conv_phased(p, phases);
      return y;
    }
    // (x.H, x.W >= 2: plan_conv then always finds a tile whose all-taps window fits one patch, which a phased launch needs;
    // an odd output width -- the mel ResUNet prunes the time axis only -- has no such view: addressed in units of C instead, below)
    if (prune_w && x.H >= 2 && x.W >= 2 && !(pb.h->cfg.tuning & VFX_TUNE_NO_FUSED_UNET)) {
      // An even output width makes (B, 2H, 2W, C) a (B, 2H, W, 2C) tensor whose channel halves are the two column parities:
      // the column classes of one row class are the PHASES of one launch (own taps, own weight tensor, one patch) -- two
      // launches per upsampler instead of four.  In the deep levels a launch is a chain of patch round trips, one per 32-channel
      // stage, whatever it computes (25-45 us each, profiles/r03_c10_ring4_ab.jsonl): half the launches, half that time.
      for (int a = 0; a < 2; ++a) {
        TapConvParams p{};
        p.B = B;
        p.Hi = x.H;
        p.Wi = x.W;
        p.Ho = y.H;
        p.Wo = x.W;
        p.Cout = 2 * D.cout;
        p.sh = 2;
        p.sw = 1;
        p.oh0 = a;
        p.ow0 = 0;
        p.Hg = (y.H - a + 1) / 2;
        p.Wg = x.W;
        p.out = const_cast<float*>(rel_ptr(y.off));
        p.act_slope = 1.f;
        p.nseg = 1;
        std::vector<TapSeg> phases(2);
        for (int b = 0; b < 2; ++b) {
          TapSeg& S = phases[b];
          S = TapSeg{};
          S.src = rel_ptr(x.off);
          S.C = x.C;
          S.scale = D.bn_scale;
          S.shift = D.bn_shift;
          S.act = ACT_LEAKY;
          S.slope = 0.f;  // ReLU
          S.wt = D.wT[a * 2 + b];
          parity_taps(S, a, b);
        }
        p.seg[0] = phases[0];  // column class 0 reads kw = 0, 2: its taps are the union of both classes' windows
        pb.add_conv_phased(p, phases);
      }
      return y;
    }
#ifndef VFX_ABL_NO_ODD_PHASED
    if (!prune_w && x.H >= 2 && x.W >= 2 && !(pb.h->cfg.tuning & VFX_TUNE_NO_FUSED_UNET)) {
      // An odd output width (2 W + 1: the mel ResUNet) has no such view, but the same two launches: the output is addressed in
      // units of C channels at the TRUE pixel index (TapConvParams::out_cmul), column class 1 one column short.
      for (int a = 0; a < 2; ++a) {
        TapConvParams p{};
        p.B = B;
        p.Hi = x.H;
        p.Wi = x.W;
        p.Ho = y.H;
        p.Wo = y.W;
        p.Cout = 2 * D.cout;
        p.out_cmul = D.cout;
        p.sh = 2;
        p.sw = 2;
        p.oh0 = a;
        p.ow0 = 0;
        p.Hg = (y.H - a + 1) / 2;
        p.Wg = x.W + 1;  // column class 0 has W + 1 columns, class 1 W
        p.out = const_cast<float*>(rel_ptr(y.off));
        p.act_slope = 1.f;
        p.nseg = 1;
        std::vector<TapSeg> phases(2);
        for (int b = 0; b < 2; ++b) {
          TapSeg& S = phases[b];
          S = TapSeg{};
          S.src = rel_ptr(x.off);
          S.C = x.C;
          S.scale = D.bn_scale;
          S.shift = D.bn_shift;
          S.act = ACT_LEAKY;
          S.slope = 0.f;  // ReLU
          S.wt = D.wT[a * 2 + b];
          parity_taps(S, a, b);
        }
        p.seg[0] = phases[0];
        pb.add_conv_phased(p, phases);
      }
      return y;
    }
#endif
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b) {
        TapConvParams p{};
        p.B = B;
        p.Hi = x.H;
        p.Wi = x.W;
        p.Ho = y.H;
        p.Wo = y.W;
        p.Cout = D.cout;
        p.sh = p.sw = 2;
        p.oh0 = a;
        p.ow0 = b;
        p.Hg = (y.H - a + 1) / 2;
        p.Wg = (y.W - b + 1) / 2;
        p.out = const_cast<float*>(rel_ptr(y.off));
        p.act_slope = 1.f;
        p.nseg = 1;
        TapSeg& S = p.seg[0];
        S.src = rel_ptr(x.off);
        S.C = x.C;
        S.scale = D.bn_scale;
        S.shift = D.bn_shift;
        S.act = ACT_LEAKY;
        S.slope = 0.f;  // ReLU
        S.wt = D.wT[a * 2 + b];
        parity_taps(S, a, b);
        pb.add_conv(p);
      }
    return y;
  }

  // Whole trunk from the single-channel input plane x1 (B, Tpad, W0) to the 32-channel tensor
  // in front of after_conv2.
  Act4 run(size_t x_off, int Tpad, int W0, bool both) {
    Plan* pl = pb.plan;
    const UNetWeights* wp = &Wt;
    const int Bc = B;
    Act4 skips[6];
    // encoder_block1.conv_block1 (Cin = 1): one launch of the fused block's entry form (resblock.hip, IN1), or -- fp32 mode,
    // VFX_TUNE_NO_FUSED_UNET / VFX_TUNE_SMALL_2D_TILES -- k_conv_c1 (conv1 and the shortcut) + k_conv (conv2)
    Act4 y;
    const int tun = pb.h->cfg.tuning;
#ifdef VFX_ABL_NO_IN1  // measurement builds (scripts/build_variant.sh)
    const bool entry_fused = false;
#else
    const bool entry_fused = !(tun & (VFX_TUNE_NO_FUSED_UNET | VFX_TUNE_SMALL_2D_TILES)) && pb.h->cfg.precision != 0;
#endif
    if (entry_fused) {
      y = make(Tpad, W0, 32);
      const ConvBlockW& w = Wt.enc[0][0];
      ResBlockParams rp{};
      rp.geo2d = 1;
      rp.in1 = 1;
      rp.x = rel_ptr(x_off);
      rp.y = const_cast<float*>(rel_ptr(y.off));
      rp.w1 = Wt.c1_w;
      rp.w2 = w.w2;
      rp.in1_scale = Wt.c1_scale;
      rp.in1_shift = Wt.c1_shift;
      rp.wsc = Wt.c1_wsc;
      rp.bsc = Wt.c1_bsc;
      rp.sc2 = w.bn2_scale;
      rp.sh2 = w.bn2_shift;
      rp.slope = kSlope;
      rp.B = B;
      rp.H = Tpad;
      rp.W = W0;
      rp.C = 32;
      pb.add_resblock(rp);
    } else {
    Act4 h1 = make(Tpad, W0, 32), sc1 = make(Tpad, W0, 32);
    {
      const size_t ho = h1.off, so = sc1.off;
      pl->ops.push_back([=](const RunCtx& c) {
        launch_conv_c1(reinterpret_cast<const float*>(pl->bound_base + x_off), Bc, Tpad, W0, wp->c1_w, wp->c1_scale,
                       wp->c1_shift, kSlope, wp->c1_wsc, wp->c1_bsc, reinterpret_cast<float*>(pl->bound_base + ho),
                       reinterpret_cast<float*>(pl->bound_base + so), c.stream);
      });
    }
    y = conv_block(Wt.enc[0][0], nullptr, 0, &h1, &sc1);
    pb.free(h1.off);
    pb.free(sc1.off);
    }
    for (int l = 0; l < 6; ++l) {
      for (int j = (l == 0 ? 1 : 0); j < 4; ++j) {
        Act4 y2 = conv_block(Wt.enc[l][j], &y, 1);
        pb.free(y.off);
        y = y2;
      }
      skips[l] = y;
      y = pool(y);
    }
    {
      Act4 y2 = conv_block(Wt.bott, &y, 1);
      pb.free(y.off);
      y = y2;
    }
    for (int d = 0; d < 6; ++d) {
      const DecoderW& D = Wt.dec[d];
      const Act4& skip = skips[5 - d];
      Act4 up = upsample(D, y, both);
      pb.free(y.off);
      VFX_CHECK(up.H == skip.H && up.W == skip.W, "decoder %d: upsampled %dx%d vs skip %dx%d", d + 1, up.H, up.W, skip.H,
                skip.W);
      Act4 srcs[2] = {up, skip};
      y = conv_block(D.blocks[0], srcs, 2);
      pb.free(up.off);
      pb.free(skip.off);
      for (int j = 1; j < 4; ++j) {
        Act4 y2 = conv_block(D.blocks[j], &y, 1);
        pb.free(y.off);
        y = y2;
      }
    }
    Act4 y2 = conv_block(Wt.after, &y, 1);
    pb.free(y.off);
    return y2;
  }
};

float* resolve(const Plan* pl, const RunCtx& c, const BufRef& b) {
  return b.ext ? c.ext[b.slot] : reinterpret_cast<float*>(pl->bound_base + b.off);
}

}  // namespace

void build_unet_mel(PlanBuilder& pb, int B, int T, BufRef mel_linear, BufRef logmel_out) {
  VFX_CHECK(pb.h->unet[VFX_MODEL_UNET_MEL], "mel ResUNet weights are not finalized");
  const UNetWeights& Wt = *pb.h->unet[VFX_MODEL_UNET_MEL];
  const int Tpad = (T + 63) / 64 * 64, W0 = 127;
  pb.short_clip = Tpad <= 128 ? 1 : (Tpad >= 2048 ? -(Tpad / 1024) : 0);  // split-K rule of the deep levels (TapConvParams::short_clip)
  Plan* pl = pb.plan;
  // a varlen batch (PlanBuilder::lens_t): every clip has its own frame count inside the SAME padded length -- the rows past it
  // are zeros like the network's own time padding (unet.py:75-77), so the trunk computes for each clip what its batch-of-one
  // call computes; no kernel of the trunk needs to know
  const int* lens_t = pb.lens_t;
  const size_t x_off = pb.alloc_f((int64_t)B * Tpad * W0);
  pl->ops.push_back([=](const RunCtx& c) {
    launch_prep_logmel(resolve(pl, c, mel_linear), B, T, Tpad, reinterpret_cast<float*>(pl->bound_base + x_off), c.flags,
                       c.stream, lens_t);
  });
  TrunkBuilder tb{pb, Wt, B};
  Act4 y = tb.run(x_off, Tpad, W0, /*both=*/false);
  pb.free(x_off);
  const UNetWeights* wp = &Wt;
  pl->ops.push_back([=](const RunCtx& c) {
    launch_final_1x1(reinterpret_cast<const float*>(pl->bound_base + y.off), B, Tpad, W0, wp->final_w, wp->final_b, 0, T,
                     resolve(pl, c, mel_linear), nullptr, resolve(pl, c, logmel_out), nullptr, c.stream);
  });
  pb.free(y.off);
}

void build_unet_spec(PlanBuilder& pb, int B, int T, BufRef sp, BufRef cosb, BufRef sinb, BufRef re_out, BufRef im_out) {
  VFX_CHECK(pb.h->unet[VFX_MODEL_UNET_SPEC], "spectrogram ResUNet weights are not finalized");
  const UNetWeights& Wt = *pb.h->unet[VFX_MODEL_UNET_SPEC];
  const int Tpad = (T + 63) / 64 * 64, W0 = 1024;
  pb.short_clip = Tpad <= 128 ? 1 : (Tpad >= 2048 ? -(Tpad / 1024) : 0);  // split-K rule of the deep levels (TapConvParams::short_clip)
  Plan* pl = pb.plan;
  const int* lens_t = pb.lens_t;  // a varlen batch: the rows past a clip's own frames are zeros (cf. build_unet_mel)
  const size_t x_off = pb.alloc_f((int64_t)B * Tpad * W0);
  pl->ops.push_back([=](const RunCtx& c) {
    launch_prep_spec(resolve(pl, c, sp), B, T, Tpad, reinterpret_cast<float*>(pl->bound_base + x_off), c.stream, lens_t);
  });
  TrunkBuilder tb{pb, Wt, B};
  Act4 y = tb.run(x_off, Tpad, W0, /*both=*/true);
  pb.free(x_off);
  const UNetWeights* wp = &Wt;
  pl->ops.push_back([=](const RunCtx& c) {
    launch_final_1x1(reinterpret_cast<const float*>(pl->bound_base + y.off), B, Tpad, W0, wp->final_w, wp->final_b, 1, T,
                     resolve(pl, c, cosb), resolve(pl, c, sinb), resolve(pl, c, re_out), resolve(pl, c, im_out), c.stream);
  });
  pb.free(y.off);
}

}  // namespace vfx
