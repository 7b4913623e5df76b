// vocoder.cpp -- weights and launch plan of the TFGAN mel -> waveform generator.
//
// Replaces `model.vocoder(mel)` (eval_gsr_voicefixer.py:66; third-party `voicefixer.Vocoder`,
// requirements.txt:6 -- structure as restated in oracle/vocoder.py, layer table in vfx_config).
// Activations are channels-last fp32 (B, T', C); the mel tensor (B,1,T,128) the reference passes
// already IS channels-last.  Every Conv1d / ConvTranspose1d with Cout >= 32 is a launch of the
// tap-convolution kernel (H = 1): ELU / LeakyReLU are prologues of the consuming convolution,
// biases and the ResStack residual adds are epilogues, a stride-s transposed convolution is s
// output-phase launches with two taps each, ReflectionPad1d is an addressing mode.
#include <cmath>
#include <cstdlib>

#include "vfx_internal.h"

namespace vfx {

int64_t vocoder_out_len(const vfx_config& cfg, int T) {
  int64_t hop = 1;
  for (int i = 0; i < cfg.voc_n_stages; ++i) hop *= cfg.voc_scales[i];
  return (int64_t)(T + T % 2 + 4) * hop;
}

namespace {

const HostTensor& staged(vfx_handle* h, const std::string& name) {
  auto& m = h->staged[VFX_MODEL_VOCODER];
  auto it = m.find(name);
  VFX_CHECK(it != m.end(), "missing vocoder tensor '%s'", name.c_str());
  return it->second;
}

// Weight packing mode (pack_conv) of a convolution of the vocoder: in the 16-bit mode a convolution that reads an
// ACTIVATED tensor (fp16, 64-channel stages) and one that reads a raw fp32 tensor (32-channel stages) use different
// fragment orders, so the mode follows the plan (build_vocoder): which tensors exist in activated form.
int pack_mode(const vfx_config& cfg, bool src_act) {
  if (cfg.precision == 2) return src_act ? 3 : 2;
  return cfg.precision != 0 ? 1 : 0;
}

// HBM-bound stacks (C = 64, 128) run each layer as ONE fused launch on the raw trunk (resblock.hip, resblock_rw.hip,
// resblock_r128.hip); the wide stacks run on a trunk in activated form (C = 256 in the 16-bit mode: one launch per layer,
// resblock_w64.hip; otherwise two k_conv launches per layer).  VFX_TUNE_NO_FUSED_STACKS forces the latter everywhere.
bool stack_fused(const vfx_config& cfg, int channels) {
  return cfg.precision != 0 && resblock_supported(channels) && !(cfg.tuning & VFX_TUNE_NO_FUSED_STACKS);
}
bool stack_fused_wide(const vfx_config& cfg, int channels) {
  return !stack_fused(cfg, channels) && cfg.precision == 2 && resblock_w64_supported(channels) && !(cfg.tuning & VFX_TUNE_NO_FUSED_WIDE);
}

// An ACTIVATED tensor of the 16-bit mode is an fp16 tensor read in 64-channel stages (pack_conv mode 3, k_conv's H64 form): it exists
// for channel counts that are multiples of 64 only.  A narrower tensor (a 32-channel stack of another layer table:
// tests/test_gpu_models.py::test_vocoder_alternate_tables) stays raw fp32 and its consumers apply their prologues themselves.
bool act_form_ok(const vfx_config& cfg, int channels) { return cfg.precision != 2 || channels % 64 == 0; }

// Packing of the fused layers' weights: pack_mode(raw source)
int fused_layer_mode(const vfx_config& cfg, int) { return pack_mode(cfg, false); }

// fp16 trunk of the 16-bit mode (round 4; VFX_TUNE_F32_TRUNK switches it off): the tensors BETWEEN the fused launches of a
// ResStack are fp16 (ResBlockParams::x16) wherever one of the 16-bit kernels runs the stack -- resblock_rw (C = 64),
// resblock_r128 (C = 128): the raw trunk as fp16; resblock_w64 (C = 256): the activated fp16 tensor alone.  A wide stack that
// feeds the vocoder tail (no upsampler behind it) keeps the two-form trunk: the tail reads raw values.
bool stack_trunk_f16(const vfx_config& cfg, int channels, bool last_stage) {
  if (cfg.precision != 2 || (cfg.tuning & VFX_TUNE_F32_TRUNK)) return false;
  if (cfg.voc_res_slope <= 0.f || cfg.voc_res_slope > 1.f) return false;  // packed max(x, slope x); the wide layer inverts the LeakyReLU
  if (stack_fused_wide(cfg, channels)) return !last_stage;
  // two k_conv launches per layer (C = 512; every stack under VFX_TUNE_NO_FUSED_*): the activated fp16 tensor alone, conv2
  // recovers its residual from it (TapConvParams::residual_act) -- 10 instead of 16 bytes per element and layer
  if (!stack_fused(cfg, channels)) return !last_stage && channels % 64 == 0;
  return channels == 128 || (channels == 64 && resblock_rw_tile(cfg.tuning) != 0);
}

// Conv1d weight (Cout, Cin, K) -> packed with taps k = 0..K-1
VocConvW load_conv1d(vfx_handle* h, const std::string& p, int cin, int cout, int K, bool src_act) {
  const HostTensor& w = staged(h, p + ".weight");
  VFX_CHECK(w.shape == std::vector<int64_t>({cout, cin, K}), "vocoder tensor '%s.weight' has an unexpected shape", p.c_str());
  std::vector<std::pair<int, int>> taps;
  for (int k = 0; k < K; ++k) taps.push_back({0, k});
  VocConvW c;
  c.cin = cin;
  c.cout = cout;
  c.mode = pack_mode(h->cfg, src_act);
  c.w = h->blob.upload(pack_conv(w.data.data(), cout, cin, 1, K, 0, cin, taps, c.mode));
  const HostTensor& b = staged(h, p + ".bias");
  VFX_CHECK((int)b.data.size() == cout, "vocoder tensor '%s.bias' has an unexpected shape", p.c_str());
  c.bias = h->blob.upload(b.data);
  return c;
}

// Taps of output phase r of ConvTranspose1d(k = 2s, stride s, padding p): out[s*q + r] +=
// x[q - e] * W[:, :, s*e + r + p] for every e with 0 <= s*e + r + p < 2s.
std::vector<std::pair<int, int>> phase_taps(int s, int pad, int r) {
  std::vector<std::pair<int, int>> t;  // (e, k)
  for (int e = -2; e <= 2; ++e) {
    const int k = s * e + r + pad;
    if (k >= 0 && k < 2 * s) t.push_back({e, k});
  }
  return t;
}

}  // namespace

std::shared_ptr<VocoderWeights> build_vocoder_weights(vfx_handle* h) {
  const vfx_config& cfg = h->cfg;
  auto W = std::make_shared<VocoderWeights>();
  f16_weight_issue() = false;  // set by pack_conv when a tensor does not fit the fp16 operands of the 16-bit mode
  int cin = cfg.n_mels;
  char name[96];
  for (int i = 0; i < cfg.voc_cond_layers; ++i) {
    snprintf(name, sizeof(name), "condnet.%d", 2 * i);
    W->cond.push_back(load_conv1d(h, name, cin, cfg.voc_cond_channels, 3, /*src_act=*/i > 0));  // the first reads the raw mel
    cin = cfg.voc_cond_channels;
  }
  W->pre = load_conv1d(h, "generator.1", cin, cfg.voc_channels, 7, /*src_act=*/true);
  int c = cfg.voc_channels, idx = 3;
  for (int st = 0; st < cfg.voc_n_stages; ++st) {
    const int s = cfg.voc_scales[st], pad = s / 2 + s % 2;
    snprintf(name, sizeof(name), "generator.%d.layer", idx);
    const HostTensor& w = staged(h, std::string(name) + ".weight");
    VFX_CHECK(w.shape == std::vector<int64_t>({c, c / 2, 2 * s}), "vocoder tensor '%s.weight' has an unexpected shape", name);
    VocConvW up;
    up.cin = c;
    up.cout = c / 2;
    // the upsampler reads the activated trunk, unless the stack in front of it is fused on the raw trunk only (in the
    // 16-bit mode the last fused layer also writes the activated fp16 form)
    up.mode = pack_mode(cfg, (st == 0 || !stack_fused(cfg, c) || cfg.precision == 2) && act_form_ok(cfg, c));
    for (int r = 0; r < s; ++r) {
      std::vector<std::pair<int, int>> taps;
      for (auto& ek : phase_taps(s, pad, r)) taps.push_back({0, ek.second});
      up.w_phase.push_back(h->blob.upload(pack_conv_transposed(w.data.data(), c, c / 2, 1, 2 * s, taps, up.mode)));
    }
    {  // the phased launch sees the output as (B, T, stride * cout): one bias copy per phase
      const std::vector<float>& b = staged(h, std::string(name) + ".bias").data;
      VFX_CHECK((int)b.size() == c / 2, "vocoder tensor '%s.bias' has an unexpected shape", name);
      std::vector<float> rep;
      for (int r = 0; r < s; ++r) rep.insert(rep.end(), b.begin(), b.end());
      up.bias = h->blob.upload(rep);
    }
    W->up.push_back(up);
    c /= 2;
    std::vector<std::pair<VocConvW, VocConvW>> stack;
    for (int i = 0; i < cfg.voc_depth[st]; ++i) {
      char a[96], b[96];
      snprintf(a, sizeof(a), "generator.%d.res_layers.%d.1", idx + 1, i);
      snprintf(b, sizeof(b), "generator.%d.res_layers.%d.3", idx + 1, i);
      // unfused layers read the activated trunk / the activated h; the fused kernel transforms raw patches itself
      const bool act = !stack_fused(cfg, c) && act_form_ok(cfg, c);  // (mode 3 = the packing for activated sources)
      stack.push_back({load_conv1d(h, a, c, c, 3, act), load_conv1d(h, b, c, c, 3, act)});
    }
    W->res.push_back(stack);
    idx += 3;
  }
  // nn.Sequential indexing: the last ResStack is followed by act (idx - 1 + ... ), ReflectionPad1d, conv:
  // after the loop idx = 3 + 3 * n_stages, the activation sits at idx - 1, the pad at idx, the conv at idx + 1.
  snprintf(name, sizeof(name), "generator.%d", idx + 1);
  const HostTensor& fw = staged(h, std::string(name) + ".weight");
  VFX_CHECK(fw.shape == std::vector<int64_t>({1, c, 7}), "final vocoder conv has an unexpected shape");
  std::vector<float> wt(7 * c);
  for (int k = 0; k < 7; ++k)
    for (int ch = 0; ch < c; ++ch) wt[k * c + ch] = fw.data[ch * 7 + k];
  W->final_w = h->blob.upload(wt);
  W->final_b = staged(h, std::string(name) + ".bias").data[0];
  W->final_c = c;
  VFX_CHECK(c % 32 == 0, "vocoder: final channel count %d must be a multiple of 32", c);
  W->needs_strict = f16_weight_issue();
  f16_weight_issue() = false;
  return W;
}

void build_vocoder(PlanBuilder& pb, int B, int T, BufRef mel_linear, BufRef wav_out, const BufRef* peak) {
  VFX_CHECK(pb.h->voc, "vocoder weights are not finalized");
  const vfx_config cfg = pb.h->cfg;
  const VocoderWeights* W = pb.h->voc.get();
  Plan* pl = pb.plan;
  vfx_handle* hh = pb.h;
  const int Tp = T + T % 2 + 4;
  // A varlen batch (PlanBuilder::lens_t / lens_tp): clip b has lens_t[b] <= T frames, i.e. lens_tp[b] <= Tp vocoder frames (its
  // OWN tail of -R frames included).  Every launch below carries lens_tp and the number of positions per vocoder frame at its
  // rate: a clip's sequence ends there (zero padding / reflection / nothing stored), as in its batch-of-one call.
  const int* lens_t = pb.lens_t;
  const int* lens_tp = pb.lens_tp;
  int rate = 1;  // positions per vocoder frame of the current stage
  pb.no_splitk = true;  // (PlanBuilder::no_splitk; reset at the end)
  pb.short_clip = 0;  // (the ResUNets' split-K rule for short clips does not apply to the vocoder's launches)

  auto resolve = [pl](const RunCtx& c, const BufRef& b) -> float* {
    return b.ext ? c.ext[b.slot] : reinterpret_cast<float*>(pl->bound_base + b.off);
  };

  size_t x = pb.alloc_f((int64_t)B * Tp * cfg.n_mels);
  {
    const size_t xo = x;
    pl->ops.push_back([=](const RunCtx& c) {
      launch_voc_prep(resolve(c, mel_linear), B, T, Tp, hh->fe.voc_inv_weight, cfg.voc_amp_floor, cfg.voc_min_db,
                      cfg.voc_norm_range, reinterpret_cast<float*>(pl->bound_base + xo), c.stream, lens_t);
      // 16-bit mode on weights that do not fit fp16 operands: every call says so (never silently wrong)
      if (hh->voc && hh->voc->needs_strict) launch_or_flags(c.flags, VFX_FLAG_F16_SATURATED, c.stream);
    });
  }

  // A tensor between two convolutions exists in up to two forms: raw fp32 (residual adds,
  // non-GEMM consumers) and ACTIVATED for its consumer convolution (that consumer's prologue applied once by the
  // producer's epilogue, MFMA operand form, staged by the DMA engine with no arithmetic).
  // floats an activated tensor of n elements occupies: fp16 (2 bytes per element) in the 16-bit mode
  auto act_floats = [&](int64_t n) -> int64_t { return cfg.precision == 2 ? (n + 1) / 2 : n; };
  constexpr size_t kNone = ~size_t(0);
  struct Forms {
    size_t raw = kNone, act = kNone;
  };
  auto free_forms = [&](const Forms& f) {
    if (f.raw != kNone) pb.free(f.raw);
    if (f.act != kNone) pb.free(f.act);
  };
  // Conv1d src -> (raw and / or activated) output.  `src_act`: src is the activated form (no prologue here);
  // `next_act` != ACT_NONE: also / only store the output activated for its consumer.
  // `residual_act`: *residual is the ACTIVATED fp16 form of the residual (LeakyReLU(res_slope)), inverted in the epilogue
  auto conv1d = [&](const VocConvW& cw, size_t src, int Tlen, int K, int dil, int act, float slope, bool reflect,
                    const size_t* residual, bool src_act, bool want_raw, int next_act, float next_slope,
                    bool residual_act = false) -> Forms {
    Forms out;
    TapConvParams p{};
    set_conv1d_geometry(p, B, Tlen, K, dil, reflect);
    p.hionly = cfg.precision == 2;
    p.Cout = cw.cout;
    p.bias = cw.bias;
    if (residual && residual_act) {
      p.residual_act = rel_ptr(*residual);
      p.residual_inv_slope = 1.f / cfg.voc_res_slope;
    } else {
      p.residual = residual ? rel_ptr(*residual) : nullptr;
    }
    p.act_slope = 1.f;
    VFX_CHECK(cw.mode == pack_mode(cfg, src_act), "vocoder plan: weights of a %d -> %d convolution are packed for another source form", cw.cin, cw.cout);
    if (want_raw) {
      out.raw = pb.alloc_f((int64_t)B * Tlen * cw.cout);
      p.out = const_cast<float*>(rel_ptr(out.raw));
    }
    if (next_act != ACT_NONE) {
      out.act = pb.alloc_f(act_floats((int64_t)B * Tlen * cw.cout));
      p.out_act = const_cast<float*>(rel_ptr(out.act));
      p.act_slope = next_slope;
      p.act_elu = next_act == ACT_ELU;
    }
    p.lens = lens_tp;
    p.lens_mul_in = p.lens_mul_out = rate;
    p.nseg = 1;
    TapSeg& S = p.seg[0];
    S.src = rel_ptr(src);
    S.C = cw.cin;
    S.act = src_act ? ACT_NONE : act;
    S.slope = src_act ? 1.f : slope;
    S.src_act = src_act ? 1 : 0;
    S.wt = cw.w;
    pb.add_conv(p);
    return out;
  };

  // condnet: Conv1d k3 + ELU; every output feeds exactly one convolution, so the ELU (and the
  // LeakyReLU in front of the first upsampler) is applied by the producer's epilogue
  Forms cur;
  cur.raw = x;
  for (size_t i = 0; i < W->cond.size(); ++i) {
    const Forms y = conv1d(W->cond[i], i > 0 ? cur.act : cur.raw, Tp, 3, 1, ACT_NONE, 1.f, false, nullptr, /*src_act=*/i > 0,
                           /*want_raw=*/false, ACT_ELU, 1.f);
    free_forms(cur);
    cur = y;
  }
  {  // ReflectionPad1d(3) + Conv1d k7 on ELU(condnet output); its output is read by upsampler 0 only
    const Forms y = conv1d(W->pre, cur.act, Tp, 7, 1, ACT_ELU, 1.f, true, nullptr, /*src_act=*/true, /*want_raw=*/false, ACT_LEAKY,
                           cfg.voc_up_slope);
    free_forms(cur);
    cur = y;
  }
  int Tlen = Tp;
  bool trunk_is_f16 = false;  // form of the current stack's raw trunk
  for (int st = 0; st < cfg.voc_n_stages; ++st) {
    const int s = cfg.voc_scales[st], pad = s / 2 + s % 2;
    const VocConvW& up = W->up[st];
    const int Tout = Tlen * s;
    const bool fuse = stack_fused(cfg, up.cout);
    // 16-bit mode, C = 256: fused as well, on the activated trunk (resblock_w64.hip); VFX_TUNE_NO_FUSED_WIDE: two launches
    const bool fuse_act = stack_fused_wide(cfg, up.cout);
    const bool last_stage = st + 1 == cfg.voc_n_stages;
    // fp16 trunk: the upsampler writes ONE fp16 tensor -- the raw trunk for resblock_rw / resblock_r128 (stored through the
    // epilogue's activated output with the identity as activation), the activated trunk for resblock_w64
    const bool t16 = stack_trunk_f16(cfg, up.cout, last_stage);
    trunk_is_f16 = t16 && fuse;
    const int64_t nel = (int64_t)B * Tout * up.cout;
    Forms y;
    if (t16 && fuse) y.raw = pb.alloc_f(act_floats(nel));
    else if (!t16) y.raw = pb.alloc_f(nel);
    const bool act_ok = act_form_ok(cfg, up.cout);  // (false: a 32-channel stack of the 16-bit mode -- raw fp32 tensors throughout)
    if (!fuse && act_ok) y.act = pb.alloc_f(act_floats(nel));
    const bool up_src_act = cur.act != kNone;  // the producer already applied LeakyReLU(up_slope)
    VFX_CHECK(up.mode == pack_mode(cfg, up_src_act), "vocoder plan: upsampler %d is packed for another source form", st);
    {
      // ConvTranspose1d(k = 2s, stride s) as ONE phased launch: output phase r (samples s*q + r) is a 2-tap
      // convolution of the input, and (B, Tlen * s, cout) viewed as (B, Tlen, s * cout) makes the phases plain cout
      // ranges -- the input patch is read from HBM once for all of them (it was read s times as s launches).
      TapConvParams p{};
      p.hionly = cfg.precision == 2;
      p.B = B;
      p.Hi = p.Hg = p.Ho = 1;
      p.Wi = p.Wg = p.Wo = Tlen;
      p.Cout = s * up.cout;
      p.sh = p.sw = 1;
      p.bias = up.bias;
      p.act_slope = 1.f;
      p.lens = lens_tp;  // (the phased launch addresses its output in INPUT positions: (B, Tlen, s * cout))
      p.lens_mul_in = p.lens_mul_out = rate;
      if (t16 && fuse) {
        p.out_act = const_cast<float*>(rel_ptr(y.raw));  // fp16(y): LeakyReLU with slope 1
      } else {
        if (!t16) p.out = const_cast<float*>(rel_ptr(y.raw));
        if (!fuse && act_ok) {
          p.out_act = const_cast<float*>(rel_ptr(y.act));
          p.act_slope = cfg.voc_res_slope;
        }
      }
      p.nseg = 1;
      std::vector<TapSeg> phases(s);
      int e_lo = 1 << 30, e_hi = -(1 << 30);
      for (int r = 0; r < s; ++r) {
        TapSeg& S = phases[r];
        S = TapSeg{};
        S.src = rel_ptr(up_src_act ? cur.act : cur.raw);
        S.C = up.cin;
        S.act = up_src_act ? ACT_NONE : ACT_LEAKY;
        S.slope = up_src_act ? 1.f : cfg.voc_up_slope;
        S.src_act = up_src_act ? 1 : 0;
        S.wt = up.w_phase[r];
        for (auto& ek : phase_taps(s, pad, r)) {
          S.dh[S.ntaps] = 0;
          S.dw[S.ntaps] = -ek.first;
          ++S.ntaps;
          e_lo = std::min(e_lo, -ek.first);
          e_hi = std::max(e_hi, -ek.first);
        }
      }
      TapSeg& U = p.seg[0];  // union of the phases' taps: fixes the patch window
      U = phases[0];
      U.ntaps = 0;
      for (int dw = e_lo; dw <= e_hi; ++dw) {
        U.dh[U.ntaps] = 0;
        U.dw[U.ntaps] = dw;
        ++U.ntaps;
      }
      pb.add_conv_phased(p, phases);
    }
    free_forms(cur);
    cur = y;
    Tlen = Tout;
    rate *= s;
    int dil = 1;
    const size_t nlayers = W->res[st].size();
    for (size_t li = 0; li < nlayers; ++li) {
      auto& layer = W->res[st][li];
      if (fuse) {
        ResBlockParams rp{};
        rp.x = rel_ptr(cur.raw);
        rp.x16 = t16 ? 1 : 0;
        rp.w1 = layer.first.w;
        rp.w2 = layer.second.w;
        rp.b1 = layer.first.bias;
        rp.b2 = layer.second.bias;
        rp.slope = cfg.voc_res_slope;
        rp.B = B;
        rp.T = Tlen;
        rp.C = up.cout;
        rp.dil = dil;
        rp.hionly = cfg.precision == 2;
        rp.tuning = cfg.tuning;
        rp.lens = lens_tp;
        rp.lens_mul = rate;
        VFX_CHECK(layer.first.mode == layer.second.mode &&
                      layer.first.mode == fused_layer_mode(cfg, up.cout),
                  "vocoder plan: the weights of a fused %d-channel layer are packed for another kernel", up.cout);
        // 16-bit mode, C = 64 ((1, 3), (9, 27)) and C = 128 ((1, 3)): two layers of small dilation as one launch -- the tensor
        // between them is never stored
        if (rp.hionly && li + 1 < nlayers && (resblock_rw_pair_ok(up.cout, dil, dil * cfg.voc_dilation_base, cfg.tuning) ||
                                              resblock_r128_pair_ok(up.cout, dil, dil * cfg.voc_dilation_base, cfg.tuning))) {
          auto& next = W->res[st][li + 1];
          dil *= cfg.voc_dilation_base;
          ++li;
          rp.dil2 = dil;
          rp.w1b = next.first.w;
          rp.w2b = next.second.w;
          rp.b1b = next.first.bias;
          rp.b2b = next.second.bias;
        }
        Forms ynew;
        const bool feeds_upsampler = cfg.precision == 2 && li + 1 == nlayers && !last_stage;
        if (!(t16 && feeds_upsampler)) {  // (fp16 trunk: the upsampler reads ya only -- the raw output is not stored at all)
          ynew.raw = pb.alloc_f(t16 ? act_floats((int64_t)B * Tlen * up.cout) : (int64_t)B * Tlen * up.cout);
          rp.y = const_cast<float*>(rel_ptr(ynew.raw));
        }
        if (feeds_upsampler) {
          // last layer in front of an upsampler: the activated fp16 form (LeakyReLU(up_slope)) for it
          ynew.act = pb.alloc_f(act_floats((int64_t)B * Tlen * up.cout));
          rp.ya = const_cast<float*>(rel_ptr(ynew.act));
          rp.act_slope = cfg.voc_up_slope;
        }
        pb.add_resblock(rp);
        free_forms(cur);
        cur = ynew;
      } else if (fuse_act) {
        // one launch per layer on the trunk in both forms: conv1 reads the activated fp16 trunk, the residual is the raw
        // one; the next trunk is written raw and -- unless it only feeds the vocoder tail -- activated for its consumer
        // (the next layer: LeakyReLU(res_slope); the next upsampler: LeakyReLU(up_slope))
        const bool last_layer = li + 1 == nlayers;
        Forms y2;
        if (!t16) y2.raw = pb.alloc_f((int64_t)B * Tlen * up.cout);
        if (!(last_layer && last_stage)) y2.act = pb.alloc_f(act_floats((int64_t)B * Tlen * up.cout));
        VFX_CHECK(layer.first.mode == 3 && layer.second.mode == 3, "vocoder plan: the fused wide layer needs fp16 64-channel weights");
        ResBlockParams rp{};
        rp.asrc = 1;
        rp.x16 = t16 ? 1 : 0;  // the activated tensor is the only form of the trunk: no x, no y
        rp.tuning = cfg.tuning;
        rp.lens = lens_tp;
        rp.lens_mul = rate;
        rp.x = t16 ? nullptr : rel_ptr(cur.raw);
        rp.xa = rel_ptr(cur.act);
        rp.y = t16 ? nullptr : const_cast<float*>(rel_ptr(y2.raw));
        rp.ya = y2.act != kNone ? const_cast<float*>(rel_ptr(y2.act)) : nullptr;
        rp.act_slope = last_layer ? cfg.voc_up_slope : cfg.voc_res_slope;
        rp.w1 = layer.first.w;
        rp.w2 = layer.second.w;
        rp.b1 = layer.first.bias;
        rp.b2 = layer.second.bias;
        rp.slope = cfg.voc_res_slope;
        rp.B = B;
        rp.T = Tlen;
        rp.C = up.cout;
        rp.dil = dil;
        rp.hionly = 1;
        pb.add_resblock(rp);
        free_forms(cur);
        cur = y2;
      } else if (!act_ok) {
        // no activated form at this width: both convolutions read raw fp32 tensors through their own LeakyReLU prologue
        const Forms hbuf = conv1d(layer.first, cur.raw, Tlen, 3, dil, ACT_LEAKY, cfg.voc_res_slope, false, nullptr,
                                  /*src_act=*/false, /*want_raw=*/true, ACT_NONE, 1.f);
        // (a following upsampler would read 32 channels: raw as well)
        const Forms y2 = conv1d(layer.second, hbuf.raw, Tlen, 3, 1, ACT_LEAKY, cfg.voc_res_slope, false, &cur.raw,
                                /*src_act=*/false, /*want_raw=*/true, ACT_NONE, 1.f);
        free_forms(hbuf);
        free_forms(cur);
        cur = y2;
      } else {
        // conv1 reads the activated trunk and writes h activated for conv2; conv2 adds the raw trunk and writes the
        // next trunk: both forms inside the stack, after its last layer only what the consumer reads (the next
        // upsampler: activated with its LeakyReLU slope; the vocoder tail: raw)
        const bool last_layer = li + 1 == nlayers;
        const Forms hbuf = conv1d(layer.first, cur.act, Tlen, 3, dil, ACT_LEAKY, cfg.voc_res_slope, false, nullptr,
                                  /*src_act=*/true, /*want_raw=*/false, ACT_LEAKY, cfg.voc_res_slope);
        // fp16 trunk (t16): no raw tensor in the stack -- conv2's residual is the activated trunk itself, inverted
        const bool want_raw = !t16 && (!last_layer || last_stage);
        const int next_act = last_layer && last_stage ? ACT_NONE : ACT_LEAKY;
        const Forms y2 = conv1d(layer.second, hbuf.act, Tlen, 3, 1, ACT_LEAKY, cfg.voc_res_slope, false, t16 ? &cur.act : &cur.raw,
                                /*src_act=*/true, want_raw, next_act, last_layer ? cfg.voc_up_slope : cfg.voc_res_slope,
                                /*residual_act=*/t16);
        free_forms(hbuf);
        free_forms(cur);
        cur = y2;
      }
      dil *= cfg.voc_dilation_base;
    }
  }
  x = cur.raw;
  VFX_CHECK(x != kNone, "vocoder plan: the tail needs the raw trunk");
  {
    const size_t xo = x;
    const int Tl = Tlen;
    const bool want_peak = peak != nullptr;
    const BufRef pk = peak ? *peak : BufRef{};
    const int tail_f16 = trunk_is_f16 ? 1 : 0;  // the last stack left its raw trunk as fp16
    pl->ops.push_back([=](const RunCtx& c) {
      launch_voc_final(reinterpret_cast<const float*>(pl->bound_base + xo), tail_f16, B, Tl, W->final_c, W->final_w, W->final_b,
                       cfg.voc_up_slope, resolve(c, wav_out), want_peak ? reinterpret_cast<unsigned*>(resolve(c, pk)) : nullptr,
                       c.stream, lens_tp, rate);
    });
  }
  free_forms(cur);
  pb.no_splitk = false;
}

}  // namespace vfx
