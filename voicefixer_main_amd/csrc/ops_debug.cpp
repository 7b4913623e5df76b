// ops_debug.cpp -- libvfx_test.so: kernel-level C entry points used by the parity tests (include/vfx_test.h).
// They pack PyTorch-layout weights on the fly, run one kernel family of libvfx.so and synchronise.  Not part of the product
// library: built into its own shared object that resolves the internals it uses from libvfx.so.
#include <cmath>

#include "vfx_internal.h"
#include "vfx_test.h"

using namespace vfx;

namespace {
struct Scratch {
  DeviceBlob blob;
};

void fill_seg(TapSeg& S, const float* x, int Cin, const float* scale, const float* shift, int act, float slope,
              DeviceBlob& blob) {
  S.src = x;
  S.C = Cin;
  std::vector<float> one(Cin, 1.f), zero(Cin, 0.f);
  S.scale = blob.upload(scale ? scale : one.data(), Cin);
  S.shift = blob.upload(shift ? shift : zero.data(), Cin);
  S.act = act;
  S.slope = slope;
}

void run_one(vfx_handle* h, TapConvParams& p, DeviceBlob& blob, hipStream_t s) {
  p.split = h->cfg.precision != 0;
  p.hionly = h->cfg.precision == 2;
  p.flags = h->d_flags;
  p.tuning = h->cfg.tuning;
  finish_params(p);
  std::vector<ConvStage> st(p.nstages);
  build_stages(p, h->d_ones, h->d_zeros, st.data());
  ConvStage* ds = static_cast<ConvStage*>(blob.alloc(st.size() * sizeof(ConvStage)));
  VFX_HIP(hipMemcpy(ds, st.data(), st.size() * sizeof(ConvStage), hipMemcpyHostToDevice));
  p.stages = ds;
  p.ksplit = choose_ksplit(p);
  if (p.ksplit > 1) p.ws = static_cast<float*>(blob.alloc((size_t)p.ksplit * p.B * p.out_img_stride * p.Cout * sizeof(float)));
  else p.ksplit = 0;
  TapConvParams* d = static_cast<TapConvParams*>(blob.alloc(sizeof(TapConvParams)));
  VFX_HIP(hipMemcpy(d, &p, sizeof(p), hipMemcpyHostToDevice));
  launch_conv(p, d, s);
  if (p.ksplit > 1) launch_splitk_reduce(p, s);
}

// A phased launch (TapConvParams::nphase: the column classes of a transposed convolution's row class as ONE launch), set up the way
// PlanBuilder::add_conv_phased / bind_plan do it: p.seg[0] = the union window, one stage table per phase.
void run_phased(vfx_handle* h, TapConvParams& p, const std::vector<TapSeg>& phases, DeviceBlob& blob, hipStream_t s) {
  p.split = h->cfg.precision != 0;
  p.hionly = h->cfg.precision == 2;
  p.flags = h->d_flags;
  p.tuning = h->cfg.tuning;
  p.nphase = (int)phases.size();
  p.cout_phase = p.Cout / p.nphase;
  finish_params(p);
  VFX_CHECK(!p.per_tap, "phased conv: the union of the phases' taps does not fit one patch");
  std::vector<ConvStage> st((size_t)p.nstages * p.nphase);
  for (int r = 0; r < p.nphase; ++r) {
    TapConvParams q = p;
    q.seg[0] = phases[r];
    build_stages(q, h->d_ones, h->d_zeros, st.data() + (size_t)r * p.nstages);
  }
  ConvStage* ds = static_cast<ConvStage*>(blob.alloc(st.size() * sizeof(ConvStage)));
  VFX_HIP(hipMemcpy(ds, st.data(), st.size() * sizeof(ConvStage), hipMemcpyHostToDevice));
  p.stages = ds;
  p.ksplit = 0;
  TapConvParams* d = static_cast<TapConvParams*>(blob.alloc(sizeof(TapConvParams)));
  VFX_HIP(hipMemcpy(d, &p, sizeof(p), hipMemcpyHostToDevice));
  launch_conv(p, d, s);
}
}  // namespace

extern "C" int vfx_op_conv(vfx_handle* h, const float* x, int B, int H, int W, int Cin, const float* weight, int Cout,
                           int kh, int kw, int dil_w, int reflect_w, const float* scale, const float* shift, int act,
                           float slope, const float* bias, const float* residual, float* y, void* stream) {
  try {
    VFX_CHECK(h && x && weight && y, "NULL argument");
    VFX_CHECK(kh * kw <= kMaxTaps, "vfx_op_conv: at most %d taps", kMaxTaps);
    DeviceGuard device_guard_(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    Scratch sc;
    std::vector<std::pair<int, int>> taps;
    TapConvParams p{};
    TapSeg& S = p.seg[0];
    for (int a = 0; a < kh; ++a)
      for (int b = 0; b < kw; ++b) {
        S.dh[taps.size()] = a - kh / 2;
        S.dw[taps.size()] = (b - kw / 2) * dil_w;
        taps.push_back({a, b});
      }
    S.ntaps = (int)taps.size();
    fill_seg(S, x, Cin, scale, shift, act, slope, sc.blob);
    S.wt = sc.blob.upload(pack_conv(weight, Cout, Cin, kh, kw, 0, Cin, taps, h->cfg.precision == 2 ? 2 : (h->cfg.precision != 0)));
    p.nseg = 1;
    p.B = B;
    p.Hi = p.Hg = p.Ho = H;
    p.Wi = p.Wg = p.Wo = W;
    p.sh = p.sw = 1;
    p.reflect_w = reflect_w;
    if (H == 1 && kh == 1 && !reflect_w) set_conv1d_geometry(p, B, W, kw, dil_w, false);  // folds wide dilations
    p.Cout = Cout;
    p.bias = bias ? sc.blob.upload(bias, Cout) : nullptr;
    p.residual = residual;
    p.out = y;
    p.act_slope = 1.f;
    run_one(h, p, sc.blob, s);
    VFX_HIP(hipStreamSynchronize(s));
  } catch (const vfx::Error&) {
    return 1;
  }
  return 0;
}

// The activated fp16 output of a fused layer must be fp16(LeakyReLU(y)) exactly: checked inside the debug entry points so that a
// test sees a mismatch as an error of the call.
static void check_activated_output(const float* y, const float* dya, size_t n, float slope, hipStream_t s) {
  VFX_HIP(hipStreamSynchronize(s));
  std::vector<float> hy(n);
  std::vector<_Float16> hya(n);
  VFX_HIP(hipMemcpy(hy.data(), y, n * sizeof(float), hipMemcpyDeviceToHost));
  VFX_HIP(hipMemcpy(hya.data(), dya, n * sizeof(_Float16), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; ++i) {
    const float v = hy[i] > 0.f ? hy[i] : hy[i] * slope;
    const _Float16 e = (_Float16)std::min(std::max(v, -65504.f), 65504.f);
    VFX_CHECK((float)e == (float)hya[i], "activated output differs from fp16(LeakyReLU(y)) at element %zu (%g vs %g)", i,
              (double)(float)hya[i], (double)(float)e);
  }
}

// ---- fp16 trunk of the 16-bit mode (ResBlockParams::x16): the debug entry points keep their fp32 interface and convert ----------
static bool handle_trunk_f16(const vfx_handle* h) { return h->cfg.precision == 2 && !(h->cfg.tuning & VFX_TUNE_F32_TRUNK); }

// device fp32 tensor -> a device fp16 tensor holding fp16(f(x)), f = LeakyReLU(slope) (slope = 1: the raw trunk)
static float* to_device_f16(DeviceBlob& blob, const float* dx, size_t n, float slope) {
  std::vector<float> hx(n);
  VFX_HIP(hipMemcpy(hx.data(), dx, n * sizeof(float), hipMemcpyDeviceToHost));
  std::vector<_Float16> hh(n);
  for (size_t i = 0; i < n; ++i) {
    const float v = hx[i] > 0.f ? hx[i] : hx[i] * slope;
    hh[i] = (_Float16)std::min(std::max(v, -65504.f), 65504.f);
  }
  void* d = blob.alloc(n * sizeof(_Float16));
  VFX_HIP(hipMemcpy(d, hh.data(), n * sizeof(_Float16), hipMemcpyHostToDevice));
  return static_cast<float*>(d);
}
static std::vector<_Float16> download_f16(const float* d, size_t n) {
  std::vector<_Float16> hh(n);
  VFX_HIP(hipMemcpy(hh.data(), d, n * sizeof(_Float16), hipMemcpyDeviceToHost));
  return hh;
}
// device fp16 tensor holding LeakyReLU(y, slope) (slope = 1: y itself) -> the caller's device fp32 tensor y
static void from_device_f16(const float* d16, float* dy, size_t n, float slope) {
  const std::vector<_Float16> hh = download_f16(d16, n);
  std::vector<float> hy(n);
  for (size_t i = 0; i < n; ++i) hy[i] = (float)hh[i] >= 0.f ? (float)hh[i] : (float)hh[i] / slope;
  VFX_HIP(hipMemcpy(dy, hy.data(), n * sizeof(float), hipMemcpyHostToDevice));
}
// Runs a planned layer (or pair) of the raw fp16 trunk twice -- y16 + ya, then ya alone (the last layer in front of an upsampler
// stores no raw output) -- checks that ya is the same bit pattern both times and that it is LeakyReLU(y16) up to the one rounding
// that separates fp16(LeakyReLU(s)) from LeakyReLU(fp16(s)), and returns y16 widened in the caller's fp32 tensor.
static void run_raw_f16_trunk(ResBlockParams rp, float* y_out, size_t n, float slope, DeviceBlob& blob, hipStream_t s) {
  float* y16 = static_cast<float*>(blob.alloc(n * sizeof(_Float16)));
  float* ya_a = static_cast<float*>(blob.alloc(n * sizeof(_Float16)));
  float* ya_b = static_cast<float*>(blob.alloc(n * sizeof(_Float16)));
  rp.x16 = 1;
  rp.act_slope = slope;
  for (int pass = 0; pass < 2; ++pass) {
    rp.y = pass == 0 ? y16 : nullptr;
    rp.ya = pass == 0 ? ya_a : ya_b;
    plan_resblock(rp);
    ResBlockParams* d = static_cast<ResBlockParams*>(blob.alloc(sizeof(ResBlockParams)));
    VFX_HIP(hipMemcpy(d, &rp, sizeof(rp), hipMemcpyHostToDevice));
    launch_resblock(rp, d, s);
  }
  VFX_HIP(hipStreamSynchronize(s));
  const std::vector<_Float16> hy = download_f16(y16, n), ha = download_f16(ya_a, n), hb = download_f16(ya_b, n);
  for (size_t i = 0; i < n; ++i) {
    VFX_CHECK(__builtin_bit_cast(unsigned short, ha[i]) == __builtin_bit_cast(unsigned short, hb[i]),
              "fp16 trunk: ya differs between the launch with and without the raw output at element %zu", i);
    const float v = (float)hy[i] > 0.f ? (float)hy[i] : (float)hy[i] * slope;
    VFX_CHECK(std::fabs((float)ha[i] - v) <= std::fabs(v) * (1.f / 512.f) + 1e-7f, "fp16 trunk: ya is not LeakyReLU(y) at element %zu (%g vs %g)", i,
              (double)(float)ha[i], (double)v);
  }
  from_device_f16(y16, y_out, n, 1.f);
}

// One ResStack layer y = x + conv2(lrelu(conv1(lrelu(x)) + b1)) + b2 on (B, T, C) tensors; weights in PyTorch
// Conv1d layout (C, C, 3) on the HOST.  fused != 0: k_resblock (C = 64 / 128, split-bf16 mode only);
// fused == 0: two k_conv launches with the ACTIVATED intermediate tensor (any C multiple of 32), which is what
// the plans use for the wide stacks -- this path also exercises the folded geometry for dil > 32.
extern "C" int vfx_op_resblock(vfx_handle* h, const float* x, int B, int T, int C, const float* w1, const float* b1,
                               const float* w2, const float* b2, int dil, float slope, int fused, float* y, void* stream) {
  try {
    VFX_CHECK(h && x && w1 && b1 && w2 && b2 && y, "NULL argument");
    VFX_CHECK(C % kKC == 0 && B > 0 && T > 0 && dil >= 1, "vfx_op_resblock: bad shape");
    DeviceGuard device_guard_(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    Scratch sc;
    const bool split = h->cfg.precision != 0;
    std::vector<std::pair<int, int>> taps = {{0, 0}, {0, 1}, {0, 2}};
    const int pmode = h->cfg.precision == 2 ? 2 : (int)split;
    // two-launch form, 16-bit mode: h travels as an activated fp16 tensor (64-channel stages) when C allows it
    const bool h_f16 = !fused && h->cfg.precision == 2 && C % 64 == 0;
    const bool h_act = !(h->cfg.precision == 2 && !h_f16);
    float* dw1 = sc.blob.upload(pack_conv(w1, C, C, 1, 3, 0, C, taps, pmode));
    float* dw2 = sc.blob.upload(pack_conv(w2, C, C, 1, 3, 0, C, taps, h_f16 ? 3 : pmode));
    float* db1 = sc.blob.upload(b1, C);
    float* db2 = sc.blob.upload(b2, C);
    const size_t nel = (size_t)B * T * C;
    const bool t16 = handle_trunk_f16(h);
    if (fused && h->cfg.precision == 2 && resblock_w64_supported(C)) {
      // wide fused layer of the 16-bit mode: conv1 reads xa = fp16(LeakyReLU(x)), built here on the host.  Two-form trunk
      // (VFX_TUNE_F32_TRUNK): x / y raw fp32 beside xa / ya; fp16 trunk: xa -> ya alone, y = LeakyReLU^-1(ya) for the caller.
      float* dxa = to_device_f16(sc.blob, x, nel, slope);
      float* dya = static_cast<float*>(sc.blob.alloc(nel * sizeof(_Float16)));
      ResBlockParams rp{};
      rp.asrc = 1;
      rp.x16 = t16 ? 1 : 0;
      rp.tuning = h->cfg.tuning;
      rp.x = t16 ? nullptr : x;
      rp.xa = dxa;
      rp.y = t16 ? nullptr : y;
      rp.ya = dya;
      rp.act_slope = slope;
      rp.w1 = sc.blob.upload(pack_conv(w1, C, C, 1, 3, 0, C, taps, 3));
      rp.w2 = sc.blob.upload(pack_conv(w2, C, C, 1, 3, 0, C, taps, 3));
      rp.b1 = db1;
      rp.b2 = db2;
      rp.slope = slope;
      rp.B = B;
      rp.T = T;
      rp.C = C;
      rp.hionly = 1;
      rp.flags = h->d_flags;
      rp.dil = dil;
      plan_resblock(rp);
      ResBlockParams* d = static_cast<ResBlockParams*>(sc.blob.alloc(sizeof(ResBlockParams)));
      VFX_HIP(hipMemcpy(d, &rp, sizeof(rp), hipMemcpyHostToDevice));
      launch_resblock(rp, d, s);
      if (t16) {
        VFX_HIP(hipStreamSynchronize(s));
        from_device_f16(dya, y, nel, slope);
      } else {
        check_activated_output(y, dya, nel, slope, s);
      }
    } else if (fused) {
      VFX_CHECK(split && resblock_supported(C), "vfx_op_resblock: the fused kernel needs precision 1 or 2 and C = 64 or 128 (precision 2: also 256)");
      ResBlockParams rp{};
      rp.x = x;
      rp.y = y;
      rp.w1 = dw1;
      rp.w2 = dw2;
      rp.b1 = db1;
      rp.b2 = db2;
      rp.slope = slope;
      rp.B = B;
      rp.T = T;
      rp.C = C;
      rp.hionly = h->cfg.precision == 2;
      rp.tuning = h->cfg.tuning;
      rp.flags = h->d_flags;
      rp.dil = dil;
      // fp16 trunk: wherever the plan would run this layer on it (vocoder.cpp stack_trunk_f16: resblock_r128, resblock_rw)
      if (rp.hionly && t16 && (C == 128 || (C == 64 && resblock_rw_tile(h->cfg.tuning) != 0))) {
        rp.x = to_device_f16(sc.blob, x, nel, 1.f);
        run_raw_f16_trunk(rp, y, nel, slope, sc.blob, s);
      } else {
        float* dya = nullptr;
        if (rp.hionly) {  // 16-bit mode: also the activated fp16 form a last layer writes for the upsampler behind it
          dya = static_cast<float*>(sc.blob.alloc(nel * sizeof(_Float16)));
          rp.ya = dya;
          rp.act_slope = slope;
        }
        plan_resblock(rp);
        ResBlockParams* d = static_cast<ResBlockParams*>(sc.blob.alloc(sizeof(ResBlockParams)));
        VFX_HIP(hipMemcpy(d, &rp, sizeof(rp), hipMemcpyHostToDevice));
        launch_resblock(rp, d, s);
        if (dya) check_activated_output(y, dya, nel, slope, s);
      }
    } else {
      float* hbuf = static_cast<float*>(sc.blob.alloc((size_t)B * T * C * sizeof(float)));
      TapConvParams p1{};
      set_conv1d_geometry(p1, B, T, 3, dil, false);
      p1.Cout = C;
      p1.nseg = 1;
      fill_seg(p1.seg[0], x, C, nullptr, nullptr, ACT_LEAKY, slope, sc.blob);
      p1.seg[0].wt = dw1;
      p1.bias = db1;
      if (h_act) {
        p1.out_act = hbuf;  // activated with conv2's prologue
        p1.act_slope = slope;
      } else {
        p1.out = hbuf;      // raw: conv2 applies the LeakyReLU itself
        p1.act_slope = 1.f;
      }
      run_one(h, p1, sc.blob, s);
      TapConvParams p2{};
      set_conv1d_geometry(p2, B, T, 3, 1, false);
      p2.Cout = C;
      p2.nseg = 1;
      fill_seg(p2.seg[0], hbuf, C, nullptr, nullptr, h_act ? ACT_NONE : ACT_LEAKY, h_act ? 1.f : slope, sc.blob);
      p2.seg[0].src_act = h_act ? 1 : 0;
      p2.seg[0].wt = dw2;
      p2.bias = db2;
      p2.residual = x;
      p2.out = y;
      p2.act_slope = 1.f;
      run_one(h, p2, sc.blob, s);
    }
    VFX_HIP(hipStreamSynchronize(s));
  } catch (const vfx::Error&) {
    return 1;
  }
  return 0;
}

// Host-only: the tile geometry plan_resblock() gives one fused ResStack layer (or layer pair, dil2 > 0) of `C` channels over
// sequences of `T` positions in precision mode `precision` -- so that the CPU test suite can check that the tiles of every
// kernel family (1-D, folded, pairs; 128- and 256-position tiles) write each output position exactly once.
// out[12] = fold, TH, W1, TWo, tiles_h, tiles_w, PW, P, tile_m, rw, patch_rows, asrc.  Needs no GPU and no handle.
extern "C" int vfx_plan_resblock_geometry_tuned(int C, int T, int dil, int dil2, int precision, int tuning, int* out);
extern "C" int vfx_plan_resblock_geometry(int C, int T, int dil, int dil2, int precision, int* out) {
  return vfx_plan_resblock_geometry_tuned(C, T, dil, dil2, precision, 0, out);
}

extern "C" int vfx_plan_resblock_geometry_tuned(int C, int T, int dil, int dil2, int precision, int tuning, int* out) {
  try {
    VFX_CHECK(out && T > 0 && dil >= 1 && dil2 >= 0, "bad argument");
    ResBlockParams rp{};
    rp.B = 1;
    rp.T = T;
    rp.C = C;
    rp.dil = dil;
    rp.dil2 = dil2;
    rp.hionly = precision == 2;
    rp.tuning = tuning;
    if (precision == 2 && resblock_w64_supported(C)) rp.asrc = 1;
    // the trunk form the vocoder plan would give this layer (vocoder.cpp stack_trunk_f16)
    rp.x16 = (precision == 2 && !(tuning & VFX_TUNE_F32_TRUNK) && (C == 256 || C == 128 || (C == 64 && resblock_rw_tile(tuning) != 0))) ? 1 : 0;
    plan_resblock(rp);
    const int v[12] = {rp.fold, rp.TH, rp.W1, rp.TWo, rp.tiles_h, rp.tiles_w, rp.PW, rp.P, rp.tile_m, rp.rw, rp.patch_rows, rp.asrc};
    for (int i = 0; i < 12; ++i) out[i] = v[i];
  } catch (const vfx::Error&) {
    return 1;
  }
  return 0;
}

extern "C" int vfx_plan_conv_geometry(int Hg, int Wg, int ntaps, const int* dh, const int* dw, int* out) {
  try {
    VFX_CHECK(out && dh && dw && Hg > 0 && Wg > 0 && ntaps >= 1 && ntaps <= kMaxTaps, "bad argument");
    static float dummy;
    TapConvParams p{};
    p.B = 1;
    p.Hi = p.Hg = p.Ho = Hg;
    p.Wi = p.Wg = p.Wo = Wg;
    p.sh = p.sw = 1;
    p.Cout = 32;
    p.split = 1;
    p.out = &dummy;  // never dereferenced: host-side planning only
    p.nseg = 1;
    p.seg[0].C = 32;
    p.seg[0].ntaps = ntaps;
    for (int t = 0; t < ntaps; ++t) {
      p.seg[0].dh[t] = dh[t];
      p.seg[0].dw[t] = dw[t];
    }
    finish_params(p);
    const int v[6] = {p.TH, p.TW, p.PW, p.P, p.per_tap, p.tiles_h * p.tiles_w};
    for (int i = 0; i < 6; ++i) out[i] = v[i];
  } catch (const vfx::Error&) {
    return 1;
  }
  return 0;
}

extern "C" int vfx_plan_block2d_geometry(int C, int H, int W, int kind, int tuning, int* out) {
  try {
    VFX_CHECK(out && H > 0 && W > 0 && kind >= 0 && kind <= 2, "bad argument");
    ResBlockParams rp{};
    rp.B = 1;
    rp.H = H;
    rp.W = W;
    rp.C = C;
    rp.geo2d = 1;
    rp.in1 = kind == 1;
    rp.two_src = kind == 2;
    rp.tuning = tuning;
    plan_block2d(rp);
    const int v[8] = {rp.TH, rp.W1, rp.TWo, rp.tiles_h, rp.tiles_w, rp.PW, rp.P, rp.tile_m};
    for (int i = 0; i < 8; ++i) out[i] = v[i];
  } catch (const vfx::Error&) {
    return 1;
  }
  return 0;
}

// Two consecutive ResStack layers (dilations dil, dil2) of the 16-bit mode at C = 64 as ONE launch (resblock_rw.hip, PAIR): the
// first layer's output never leaves the CU.  Weights / biases in PyTorch layout on the HOST: wa1, ba1, wa2, ba2 = first layer,
// wb1 .. bb2 = second layer.  Fails (returns 1) where the plan would not pair the layers.
extern "C" int vfx_op_resblock_pair(vfx_handle* h, const float* x, int B, int T, int C, const float* wa1, const float* ba1,
                                    const float* wa2, const float* ba2, int dil, const float* wb1, const float* bb1,
                                    const float* wb2, const float* bb2, int dil2, float slope, float* y, void* stream) {
  try {
    VFX_CHECK(h && x && y && wa1 && ba1 && wa2 && ba2 && wb1 && bb1 && wb2 && bb2 && B > 0 && T > 0, "bad argument");
    DeviceGuard device_guard_(h->device);
    VFX_CHECK(h->cfg.precision == 2 && (resblock_rw_pair_ok(C, dil, dil2, h->cfg.tuning) || resblock_r128_pair_ok(C, dil, dil2, h->cfg.tuning)),
              "vfx_op_resblock_pair: needs the 16-bit mode and a pair the plan would build (C = 64: dil <= 32, dil2 <= 62; C = 128: (1, 3))");
    hipStream_t s = static_cast<hipStream_t>(stream);
    Scratch sc;
    std::vector<std::pair<int, int>> taps = {{0, 0}, {0, 1}, {0, 2}};
    ResBlockParams rp{};
    rp.x = x;
    rp.y = y;
    rp.w1 = sc.blob.upload(pack_conv(wa1, C, C, 1, 3, 0, C, taps, 2));
    rp.w2 = sc.blob.upload(pack_conv(wa2, C, C, 1, 3, 0, C, taps, 2));
    rp.b1 = sc.blob.upload(ba1, C);
    rp.b2 = sc.blob.upload(ba2, C);
    rp.w1b = sc.blob.upload(pack_conv(wb1, C, C, 1, 3, 0, C, taps, 2));
    rp.w2b = sc.blob.upload(pack_conv(wb2, C, C, 1, 3, 0, C, taps, 2));
    rp.b1b = sc.blob.upload(bb1, C);
    rp.b2b = sc.blob.upload(bb2, C);
    rp.slope = slope;
    rp.B = B;
    rp.T = T;
    rp.C = C;
    rp.hionly = 1;
    rp.flags = h->d_flags;
    rp.dil = dil;
    rp.dil2 = dil2;
    rp.tuning = h->cfg.tuning;
    const size_t nel = (size_t)B * T * C;
    if (handle_trunk_f16(h)) {
      rp.x = to_device_f16(sc.blob, x, nel, 1.f);
      run_raw_f16_trunk(rp, y, nel, slope, sc.blob, s);
    } else {
      float* dya = static_cast<float*>(sc.blob.alloc(nel * sizeof(_Float16)));
      rp.ya = dya;
      rp.act_slope = slope;
      plan_resblock(rp);
      ResBlockParams* d = static_cast<ResBlockParams*>(sc.blob.alloc(sizeof(ResBlockParams)));
      VFX_HIP(hipMemcpy(d, &rp, sizeof(rp), hipMemcpyHostToDevice));
      launch_resblock(rp, d, s);
      check_activated_output(y, dya, nel, slope, s);
    }
  } catch (const vfx::Error&) {
    return 1;
  }
  return 0;
}

// One fused 2-D ConvBlockRes (Cin == Cout = C in {32, 64}, identity shortcut; resblock.hip, G2 mode):
//   y = x + conv2(lrelu(bn2(conv1(lrelu(bn1(x))))));  x, y (B, H, W, C) on the device; w1, w2 in PyTorch layout (C, C, 3, 3)
//   and the folded BatchNorm affines sc*/sh* [C] on the HOST.
extern "C" int vfx_op_block2d(vfx_handle* h, const float* x, int B, int H, int W, int C, const float* w1, const float* sc1,
                              const float* sh1, const float* w2, const float* sc2, const float* sh2, float slope, float* y,
                              void* stream) {
  try {
  VFX_CHECK(h && x && y && w1 && w2 && sc1 && sh1 && sc2 && sh2 && B > 0 && H > 0 && W > 0, "bad argument");
  DeviceGuard device_guard_(h->device);
  VFX_CHECK(h->cfg.precision != 0 && block2d_supported(C), "vfx_op_block2d: needs a split-bf16 mode and C = 32 or 64");
  hipStream_t s = static_cast<hipStream_t>(stream);
  Scratch sc;
  std::vector<std::pair<int, int>> taps;
  for (int kh = 0; kh < 3; ++kh)
    for (int kw = 0; kw < 3; ++kw) taps.push_back({kh, kw});
  ResBlockParams rp{};
  rp.x = x;
  rp.y = y;
  rp.w1 = sc.blob.upload(pack_conv(w1, C, C, 3, 3, 0, C, taps, 1));
  rp.w2 = sc.blob.upload(pack_conv(w2, C, C, 3, 3, 0, C, taps, 1));
  rp.sc1 = sc.blob.upload(sc1, C);
  rp.sh1 = sc.blob.upload(sh1, C);
  rp.sc2 = sc.blob.upload(sc2, C);
  rp.sh2 = sc.blob.upload(sh2, C);
  rp.slope = slope;
  rp.B = B;
  rp.H = H;
  rp.W = W;
  rp.C = C;
  rp.tuning = h->cfg.tuning;
  plan_block2d(rp);
  ResBlockParams* d = static_cast<ResBlockParams*>(sc.blob.alloc(sizeof(ResBlockParams)));
  VFX_HIP(hipMemcpy(d, &rp, sizeof(rp), hipMemcpyHostToDevice));
  launch_resblock(rp, d, s);
  if (const char* reps = getenv("VFX_OP_REPS")) {  // measurement aid: the same launch N more times between two events, average on stderr
    const int n = atoi(reps);
    hipEvent_t e0, e1;
    VFX_HIP(hipEventCreate(&e0));
    VFX_HIP(hipEventCreate(&e1));
    VFX_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < n; ++i) launch_resblock(rp, d, s);
    VFX_HIP(hipEventRecord(e1, s));
    VFX_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    VFX_HIP(hipEventElapsedTime(&ms, e0, e1));
    fprintf(stderr, "[vfx_op_block2d] %d launches: %.2f us each\n", n, 1000.0 * ms / (n > 0 ? n : 1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
  VFX_HIP(hipStreamSynchronize(s));
  } catch (const vfx::Error&) {
    return 1;
  }
  return 0;
}

extern "C" int vfx_op_conv_transpose(vfx_handle* h, const float* x, int B, int H, int W, int Cin, const float* weight,
                                     int Cout, int kh, int kw, int stride, int prune_w, const float* scale,
                                     const float* shift, int act, float slope, const float* bias, float* y,
                                     void* stream) {
  try {
    VFX_CHECK(h && x && weight && y, "NULL argument");
    DeviceGuard device_guard_(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    Scratch sc;
    const float* dbias = bias ? sc.blob.upload(bias, Cout) : nullptr;
    if (kh == 3 && kw == 3 && stride == 2 && !bias && H >= 2 && W >= 2 && Cout % 32 == 0 && !(h->cfg.tuning & VFX_TUNE_NO_FUSED_UNET)) {
      // the product's form (resunet.cpp, TrunkBuilder::upsample): all four parity classes as the PHASES of one launch, the output
      // addressed in units of Cout (out_cmul, phase_rows); VFX_TUNE_TWO_LAUNCH_UPSAMPLERS: the two column classes of a row class as
      // the phases of one launch -- even output width: the output viewed as (B, 2H, W, 2 Cout); odd width: out_cmul
      const int Ho = 2 * H, Wo = prune_w ? 2 * W : 2 * W + 1;
      const bool four = !(h->cfg.tuning & VFX_TUNE_TWO_LAUNCH_UPSAMPLERS);  // all four phases in one launch (either width)
      if (four) {
        TapConvParams p{};
        p.B = B;
        p.Hi = H;
        p.Wi = W;
        p.Ho = Ho;
        p.Wo = Wo;
        p.Cout = 4 * Cout;
        p.out_cmul = Cout;
        p.phase_rows = 1;
        p.sh = 2;
        p.sw = 2;
        p.Hg = (Ho + 1) / 2;
        p.Wg = (Wo + 1) / 2;
        p.out = y;
        p.nseg = 1;
        std::vector<TapSeg> phases(4);
        for (int ph = 0; ph < 4; ++ph) {
          TapSeg& S = phases[ph];
          S = TapSeg{};
          std::vector<std::pair<int, int>> taps;
          for (int r = ph >> 1; r < 3; r += 2)
            for (int c = ph & 1; c < 3; c += 2) {
              S.dh[taps.size()] = -(r / 2);
              S.dw[taps.size()] = -(c / 2);
              taps.push_back({r, c});
            }
          S.ntaps = (int)taps.size();
          fill_seg(S, x, Cin, scale, shift, act, slope, sc.blob);
          S.wt = sc.blob.upload(pack_conv_transposed(weight, Cin, Cout, 3, 3, taps, h->cfg.precision == 2 ? 2 : (h->cfg.precision != 0)));
        }
        p.seg[0] = phases[0];
        run_phased(h, p, phases, sc.blob, s);
      }
      for (int a = 0; a < 2 && !four; ++a) {
        TapConvParams p{};
        p.B = B;
        p.Hi = H;
        p.Wi = W;
        p.Ho = Ho;
        p.Cout = 2 * Cout;
        p.sh = 2;
        p.oh0 = a;
        p.ow0 = 0;
        p.Hg = (Ho - a + 1) / 2;
        if (prune_w) {
          p.Wo = W;
          p.sw = 1;
          p.Wg = W;
        } else {
          p.Wo = Wo;
          p.sw = 2;
          p.Wg = W + 1;
          p.out_cmul = Cout;
        }
        p.out = y;
        p.nseg = 1;
        std::vector<TapSeg> phases(2);
        for (int b = 0; b < 2; ++b) {
          TapSeg& S = phases[b];
          S = TapSeg{};
          std::vector<std::pair<int, int>> taps;
          for (int r = a; r < 3; r += 2)
            for (int c = b; c < 3; c += 2) {
              S.dh[taps.size()] = -(r / 2);
              S.dw[taps.size()] = -(c / 2);
              taps.push_back({r, c});
            }
          S.ntaps = (int)taps.size();
          fill_seg(S, x, Cin, scale, shift, act, slope, sc.blob);
          S.wt = sc.blob.upload(pack_conv_transposed(weight, Cin, Cout, 3, 3, taps, h->cfg.precision == 2 ? 2 : (h->cfg.precision != 0)));
        }
        p.seg[0] = phases[0];
        run_phased(h, p, phases, sc.blob, s);
      }
    } else if (kh == 3 && kw == 3 && stride == 2) {
      const int Ho = 2 * H, Wo = prune_w ? 2 * W : 2 * W + 1;
      for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
          TapConvParams p{};
          TapSeg& S = p.seg[0];
          std::vector<std::pair<int, int>> taps;
          for (int r = a; r < 3; r += 2)
            for (int c = b; c < 3; c += 2) {
              S.dh[taps.size()] = -(r / 2);
              S.dw[taps.size()] = -(c / 2);
              taps.push_back({r, c});
            }
          S.ntaps = (int)taps.size();
          fill_seg(S, x, Cin, scale, shift, act, slope, sc.blob);
          S.wt = sc.blob.upload(pack_conv_transposed(weight, Cin, Cout, 3, 3, taps, h->cfg.precision == 2 ? 2 : (h->cfg.precision != 0)));
          p.nseg = 1;
          p.B = B;
          p.Hi = H;
          p.Wi = W;
          p.Ho = Ho;
          p.Wo = Wo;
          p.Cout = Cout;
          p.sh = p.sw = 2;
          p.oh0 = a;
          p.ow0 = b;
          p.Hg = (Ho - a + 1) / 2;
          p.Wg = (Wo - b + 1) / 2;
          p.bias = dbias;
          p.out = y;
          run_one(h, p, sc.blob, s);
        }
    } else {
      VFX_CHECK(kh == 1 && H == 1 && kw == 2 * stride, "vfx_op_conv_transpose: unsupported geometry");
      const int sN = stride, pad = sN / 2 + sN % 2;
      for (int r = 0; r < sN; ++r) {
        TapConvParams p{};
        TapSeg& S = p.seg[0];
        std::vector<std::pair<int, int>> taps;
        for (int e = -2; e <= 2; ++e) {
          const int k = sN * e + r + pad;
          if (k >= 0 && k < 2 * sN) {
            S.dh[taps.size()] = 0;
            S.dw[taps.size()] = -e;
            taps.push_back({0, k});
          }
        }
        S.ntaps = (int)taps.size();
        fill_seg(S, x, Cin, scale, shift, act, slope, sc.blob);
        S.wt = sc.blob.upload(pack_conv_transposed(weight, Cin, Cout, 1, kw, taps, h->cfg.precision == 2 ? 2 : (h->cfg.precision != 0)));
        p.nseg = 1;
        p.B = B;
        p.Hi = p.Hg = p.Ho = 1;
        p.Wi = p.Wg = W;
        p.Wo = W * sN;
        p.Cout = Cout;
        p.sh = 1;
        p.sw = sN;
        p.ow0 = r;
        p.bias = dbias;
        p.out = y;
        run_one(h, p, sc.blob, s);
      }
    }
    VFX_HIP(hipStreamSynchronize(s));
  } catch (const vfx::Error&) {
    return 1;
  }
  return 0;
}
