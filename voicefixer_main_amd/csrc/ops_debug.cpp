// ops_debug.cpp -- kernel-level C entry points used by the parity tests (vfx_op_*).
// They pack PyTorch-layout weights on the fly, run one tap-convolution and synchronise.
#include "vfx_internal.h"

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
  finish_params(p);
  std::vector<ConvStage> st(p.nstages);
  build_stages(p, h->d_ones, h->d_zeros, st.data());
  ConvStage* ds = static_cast<ConvStage*>(blob.alloc(st.size() * sizeof(ConvStage)));
  VFX_HIP(hipMemcpy(ds, st.data(), st.size() * sizeof(ConvStage), hipMemcpyHostToDevice));
  p.stages = ds;
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
    VFX_HIP(hipSetDevice(h->device));
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
    S.wt = sc.blob.upload(pack_conv(weight, Cout, Cin, kh, kw, 0, Cin, taps, h->cfg.precision != 0));
    p.nseg = 1;
    p.B = B;
    p.Hi = p.Hg = p.Ho = H;
    p.Wi = p.Wg = p.Wo = W;
    p.Cout = Cout;
    p.sh = p.sw = 1;
    p.reflect_w = reflect_w;
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

extern "C" int vfx_op_conv_transpose(vfx_handle* h, const float* x, int B, int H, int W, int Cin, const float* weight,
                                     int Cout, int kh, int kw, int stride, int prune_w, const float* scale,
                                     const float* shift, int act, float slope, const float* bias, float* y,
                                     void* stream) {
  try {
    VFX_CHECK(h && x && weight && y, "NULL argument");
    VFX_HIP(hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    Scratch sc;
    const float* dbias = bias ? sc.blob.upload(bias, Cout) : nullptr;
    if (kh == 3 && kw == 3 && stride == 2) {
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
          S.wt = sc.blob.upload(pack_conv_transposed(weight, Cin, Cout, 3, 3, taps, h->cfg.precision != 0));
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
        S.wt = sc.blob.upload(pack_conv_transposed(weight, Cin, Cout, 1, kw, taps, h->cfg.precision != 0));
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
