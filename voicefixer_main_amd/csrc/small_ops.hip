// small_ops.hip -- HBM-bound edge kernels around the tap-convolution engine (gfx950).
// All activations are channels-last fp32; every kernel is written for 16-byte-per-lane
// coalesced access along the channel axis (or along the innermost spatial axis when C == 1).
#include "conv_common.h"
#include "vfx_internal.h"

namespace vfx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline unsigned nblocks(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

// ---------------------------------------------------------------------------------------------
// UNet input preparation
// ---------------------------------------------------------------------------------------------
// Generator.forward's to_log (models/gsr_voicefixer.py:87; pytorch_util.py:157-159) fused with
// the time padding and last-bin drop of unet.py:75-78:  (B,T,128) -> (B,Tpad,127).
// lens_t != nullptr (batches of clips of unequal length): clip b has lens_t[b] <= T frames in its T rows of `mel`; the rows past
// them are the network's zero time padding, like the rows T .. Tpad
__global__ void k_prep_logmel(const float* __restrict__ mel, int B, int T, int Tpad, float* __restrict__ x,
                              int* __restrict__ flags, const int* __restrict__ lens_t) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)B * Tpad * 127;
  const bool live = idx < total;
  const int f = idx % 127;
  const int64_t r = idx / 127;
  const int i = r % Tpad;
  const int b = r / Tpad;
  float v = 0.f;
  bool neg = false;
  if (live && i < (lens_t ? min(T, lens_t[b]) : T)) {
    const float* row = mel + ((int64_t)b * T + i) * 128;
    const float m = row[f];
    // to_log asserts on the WHOLE tensor (pytorch_util.py:158): bin 127 is dropped from the network's input but not from
    // the check -- the thread of bin 126 looks at it as well
    neg = m < 0.f || (f == 126 && row[127] < 0.f);
    v = log10f(fmaxf(m, 1e-8f));
  }
  if (live) x[idx] = v;
  if (__any(neg) && (threadIdx.x & 63) == 0) or_flag_global(flags, VFX_FLAG_NEGATIVE_INPUT);   // one global atomic per wave
}

// unet_v2.py:103-110: (B,T,1025) -> (B,Tpad,1024), zero rows beyond T, last bin dropped.
__global__ void k_prep_spec(const float* __restrict__ sp, int B, int T, int Tpad, float* __restrict__ x,
                            const int* __restrict__ lens_t /* frames per clip of a varlen batch, or null */) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)B * Tpad * 1024;
  if (idx >= total) return;
  const int f = idx & 1023;
  const int64_t r = idx >> 10;
  const int i = r % Tpad;
  const int b = r / Tpad;
  x[idx] = i < (lens_t ? min(T, lens_t[b]) : T) ? sp[((int64_t)b * T + i) * 1025 + f] : 0.f;
}

void launch_prep_logmel(const float* mel, int B, int T, int Tpad, float* x, int* flags, hipStream_t s, const int* lens_t) {
  hipLaunchKernelGGL(k_prep_logmel, dim3(nblocks((int64_t)B * Tpad * 127, 256)), dim3(256), 0, s, mel, B, T, Tpad, x, flags, lens_t);
  VFX_HIP(hipGetLastError());
}
void launch_prep_spec(const float* sp, int B, int T, int Tpad, float* x, hipStream_t s, const int* lens_t) {
  hipLaunchKernelGGL(k_prep_spec, dim3(nblocks((int64_t)B * Tpad * 1024, 256)), dim3(256), 0, s, sp, B, T, Tpad, x, lens_t);
  VFX_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// First residual block, Cin = 1 (encoder_block1.conv_block1 of modules.py:223-271):
//   h  = conv3x3( lrelu(scale*x + shift) )   zero halo AFTER the activation
//   sc = shortcut_w * x + shortcut_b          (1x1 conv with bias on the raw input)
// 8 lanes per pixel, 4 output channels each -> 16-byte coalesced stores.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_conv_c1(const float* __restrict__ x, int B, int H, int W,
                                                  const float* __restrict__ w9x32, float scale, float shift,
                                                  float slope, const float* __restrict__ wsc,
                                                  const float* __restrict__ bsc, float* __restrict__ h,
                                                  float* __restrict__ sc) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t pix = gid >> 3;
  const int cg = gid & 7;
  if (pix >= (int64_t)B * H * W) return;
  const int j = pix % W;
  const int64_t r = pix / W;
  const int i = r % H;
  const float* xb = x + (r - i) * W;  // start of image b
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int ii = i + kh - 1, jj = j + kw - 1;
      float a = 0.f;
      if (ii >= 0 && ii < H && jj >= 0 && jj < W) {
        a = xb[(int64_t)ii * W + jj] * scale + shift;
        a = a >= 0.f ? a : a * slope;
      }
      const f32x4 w = *reinterpret_cast<const f32x4*>(w9x32 + (kh * 3 + kw) * 32 + 4 * cg);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = fmaf(a, w[e], acc[e]);
    }
  }
  *reinterpret_cast<f32x4*>(h + pix * 32 + 4 * cg) = acc;
  const float xv = xb[(int64_t)i * W + j];
  const f32x4 ws = *reinterpret_cast<const f32x4*>(wsc + 4 * cg);
  const f32x4 bs = *reinterpret_cast<const f32x4*>(bsc + 4 * cg);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = fmaf(xv, ws[e], bs[e]);
  *reinterpret_cast<f32x4*>(sc + pix * 32 + 4 * cg) = o;
}

void launch_conv_c1(const float* x, int B, int H, int W, const float* w9x32, float scale, float shift, float slope,
                    const float* wsc32, const float* bsc32, float* h, float* sc, hipStream_t s) {
  hipLaunchKernelGGL(k_conv_c1, dim3(nblocks((int64_t)B * H * W * 8, 256)), dim3(256), 0, s, x, B, H, W, w9x32, scale,
                     shift, slope, wsc32, bsc32, h, sc);
  VFX_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// F.avg_pool2d(kernel 2) with floor semantics (modules.py:183): (B,H,W,C) -> (B,H/2,W/2,C)
// ---------------------------------------------------------------------------------------------
__global__ void k_avgpool2(const float* __restrict__ x, int H, int W, int C4, int Ho, int Wo, int64_t total,
                           float* __restrict__ y) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = idx % C4;
  int64_t r = idx / C4;
  const int j = r % Wo;
  r /= Wo;
  const int i = r % Ho;
  const int64_t b = r / Ho;
  const f32x4* X = reinterpret_cast<const f32x4*>(x);
  const int64_t p00 = ((b * H + 2 * i) * W + 2 * j) * C4 + c;
  const f32x4 v = ((X[p00] + X[p00 + C4]) + X[p00 + (int64_t)W * C4]) + X[p00 + (int64_t)W * C4 + C4];
  reinterpret_cast<f32x4*>(y)[idx] = v * 0.25f;
}

void launch_avgpool2(const float* x, int B, int H, int W, int C, float* y, hipStream_t s) {
  const int Ho = H / 2, Wo = W / 2, C4 = C / 4;
  const int64_t total = (int64_t)B * Ho * Wo * C4;
  hipLaunchKernelGGL(k_avgpool2, dim3(nblocks(total, 256)), dim3(256), 0, s, x, H, W, C4, Ho, Wo, total, y);
  VFX_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// after_conv2 (1x1, 32 -> 1, bias; unet.py:52-53,96) + "recover shape" (unet.py:99-100) fused
// with the caller's epilogue.  8 lanes per pixel, float4 dot + 3 xor-shuffles.
//   mode 0: out0[b,t,f] = (f < 127 ? v : 0) + log10(max(aux0[b,t,f], 1e-8))         (B,T,128)
//           = Generator.forward's  out['mel'] + to_log(mel_orig)  (gsr_voicefixer.py:90)
//   mode 1: mag = (f < 1024 ? v : 0); out0 = mag * aux0 (cos), out1 = mag * aux1 (sin)  (B,T,1025)
//           = unet_v2.py:129-137
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_final_1x1(const float* __restrict__ y, int B, int Tpad, int W,
                                                    const float* __restrict__ w32, float bias, int mode, int T,
                                                    const float* __restrict__ aux0, const float* __restrict__ aux1,
                                                    float* __restrict__ out0, float* __restrict__ out1) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t q = gid >> 3;  // pixel among B*T*W (only rows < T)
  const int cg = gid & 7;
  const bool active = q < (int64_t)B * T * W;
  float v = 0.f;
  int f = 0, t = 0;
  int64_t b = 0;
  if (active) {
    f = q % W;
    const int64_t r = q / W;
    t = r % T;
    b = r / T;
    const f32x4 a = *reinterpret_cast<const f32x4*>(y + (((b * Tpad + t) * W) + f) * 32 + 4 * cg);
    const f32x4 w = *reinterpret_cast<const f32x4*>(w32 + 4 * cg);
    v = a[0] * w[0] + a[1] * w[1] + a[2] * w[2] + a[3] * w[3];
  }
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  if (!active || cg != 0) return;
  v += bias;
  const int Fo = W + 1;
  const int64_t o = (b * T + t) * Fo + f;
  if (mode == 0) {
    out0[o] = v + log10f(fmaxf(aux0[o], 1e-8f));
    if (f == W - 1) out0[o + 1] = log10f(fmaxf(aux0[o + 1], 1e-8f));
  } else {
    out0[o] = v * aux0[o];
    out1[o] = v * aux1[o];
    if (f == W - 1) {
      out0[o + 1] = 0.f;
      out1[o + 1] = 0.f;
    }
  }
}

void launch_final_1x1(const float* y, int B, int Tpad, int W, const float* w32, float bias, int mode, int T,
                      const float* aux0, const float* aux1, float* out0, float* out1, hipStream_t s) {
  hipLaunchKernelGGL(k_final_1x1, dim3(nblocks((int64_t)B * T * W * 8, 256)), dim3(256), 0, s, y, B, Tpad, W, w32, bias,
                     mode, T, aux0, aux1, out0, out1);
  VFX_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// Vocoder input normalisation (voicefixer.Vocoder.__call__, see oracle/vocoder.py):
//   (B,T,128) linear mel -> (B,Tp,128) conditioning in [-R, R], frames >= T filled with -R.
// ---------------------------------------------------------------------------------------------
// lens_t != nullptr (batches of clips of unequal length): clip b has lens_t[b] <= T frames; its tail of -R frames follows THEM
// (rows past its own T_b + T_b % 2 + 4 are never read: the vocoder's launches stop at the clip's length)
__global__ void k_voc_prep(const float* __restrict__ mel, int T, int Tp, const float* __restrict__ inv_w,
                           float amp_floor, float min_db, float range, int64_t total, float* __restrict__ cond,
                           const int* __restrict__ lens_t) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int f = idx & 127;
  const int64_t r = idx >> 7;
  const int t = r % Tp;
  const int64_t b = r / Tp;
  float v = -range;
  if (t < (lens_t ? min(T, lens_t[b]) : T)) {
    const float m = fabsf(mel[(b * T + t) * 128 + f] * inv_w[f]);
    float s = 20.f * log10f(fmaxf(m, amp_floor)) - 20.f;
    s = (s - min_db) / (-min_db) * (2.f * range) - range;
    v = fminf(fmaxf(s, -range), range);
  }
  cond[idx] = v;
}

void launch_voc_prep(const float* mel, int B, int T, int Tp, const float* inv_weight, float amp_floor, float min_db,
                     float range, float* cond, hipStream_t s, const int* lens_t) {
  const int64_t total = (int64_t)B * Tp * 128;
  hipLaunchKernelGGL(k_voc_prep, dim3(nblocks(total, 256)), dim3(256), 0, s, mel, T, Tp, inv_weight, amp_floor, min_db,
                     range, total, cond, lens_t);
  VFX_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// Vocoder tail: LeakyReLU(slope) -> ReflectionPad1d(3) -> Conv1d(C -> 1, k7) -> tanh, plus the per-clip
// peak |wave| the handler's peak normalisation needs (eval_gsr_voicefixer.py:68-70), so that the
// waveform is not read a second time.
// HBM-bound (C * 4 bytes in, 4 bytes out per sample).  A group of 8 lanes owns 8 consecutive output
// samples and C/8 channels per lane: the 14 input rows the 8 outputs touch are loaded once (16 bytes per
// lane and row piece), the 7 x C/8 taps of the lane sit in registers, and the 8 partial sums are
// completed with three DPP-sized xor shuffles each.  One atomicMax per block for the peak.
// OPG = outputs per group of 8 lanes: 8, or 16 on the fp16 trunk (round 6: 22 rows fetched for 16 outputs instead of 14 for 8 -- 1.375
// instead of 1.75 times the tensor through the vector memory path; the same sums in the same order).
// ---------------------------------------------------------------------------------------------
template <int CPL, bool X16, int OPG = 8>  // channels per lane = C / 8 (4, 8 or 16); X16: x is the fp16 trunk of the 16-bit mode
__global__ __launch_bounds__(256) void k_voc_final(const float* __restrict__ x, int T, const float* __restrict__ w /*[7][C]*/,
                                                    float bias, float slope, float* __restrict__ wav,
                                                    unsigned* __restrict__ peak /*[B] or null*/,
                                                    const int* __restrict__ lens /*[B] vocoder frames per clip, or null*/, int hop) {
  constexpr int C = 8 * CPL;
  const int b = blockIdx.y;
  // batches of clips of unequal length: clip b ends (reflects, stops writing, stops counting for the peak) at its own length;
  // Ts stays the stride between clips
  const int Ts = T;
  if (lens) T = min(T, lens[b] * hop);
  const int grp = threadIdx.x >> 3, g = threadIdx.x & 7;
  const int t0 = (blockIdx.x * 32 + grp) * OPG;  // first output sample of the group
  const float* xb = x + (int64_t)b * Ts * C + g * CPL;
  const _Float16* xh = reinterpret_cast<const _Float16*>(x) + (int64_t)b * Ts * C + g * CPL;
  float wk[7][CPL];
#pragma unroll
  for (int k = 0; k < 7; ++k)
#pragma unroll
    for (int c = 0; c < CPL; c += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(w + k * C + g * CPL + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) wk[k][c + e] = v[e];
    }
  float acc[OPG];
#pragma unroll
  for (int o = 0; o < OPG; ++o) acc[o] = 0.f;
  if (t0 < T) {
#pragma unroll
    for (int r = 0; r < OPG + 6; ++r) {  // input row t0 - 3 + r feeds output o with tap k = r - o
      int tt = t0 - 3 + r;
      tt = tt < 0 ? -tt : tt;
      tt = tt >= T ? 2 * (T - 1) - tt : tt;
      tt = tt < 0 ? 0 : tt;  // only rows of masked outputs (t >= T) can get here
      float v[CPL];
#pragma unroll
      for (int c = 0; c < CPL; c += 4) {
        f32x4 a;
        if constexpr (X16) {
          typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
          const f16x4_t hv = *reinterpret_cast<const f16x4_t*>(xh + (int64_t)tt * C + c);
          a = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
        } else {
          a = *reinterpret_cast<const f32x4*>(xb + (int64_t)tt * C + c);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[c + e] = a[e] >= 0.f ? a[e] : a[e] * slope;
      }
#pragma unroll
      for (int o = 0; o < OPG; ++o) {
        const int k = r - o;
        if (k >= 0 && k < 7) {
#pragma unroll
          for (int c = 0; c < CPL; ++c) acc[o] = fmaf(v[c], wk[k][c], acc[o]);
        }
      }
    }
  }
  float mine[OPG / 8];  // lane g finishes outputs t0 + g (+ 8)
#pragma unroll
  for (int h = 0; h < OPG / 8; ++h) mine[h] = 0.f;
#pragma unroll
  for (int o = 0; o < OPG; ++o) {
    float s = acc[o];
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    mine[o >> 3] = g == (o & 7) ? s : mine[o >> 3];
  }
  float m = 0.f;
#pragma unroll
  for (int h = 0; h < OPG / 8; ++h) {
    const int t = t0 + 8 * h + g;
    if (t < T) {
      const float y = tanhf(mine[h] + bias);
      wav[(int64_t)b * Ts + t] = y;
      m = fmaxf(m, fabsf(y));
    }
  }
  if (peak) {
    __shared__ float wmax[4];
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
      if (m > 0.f) atomicMax(peak + b, __float_as_uint(m));  // |y| >= 0: the uint order is the float order
    }
  }
}

void launch_voc_final(const float* x, int x_f16, int B, int T, int C, const float* w, float bias, float slope, float* wav,
                      unsigned* peak, hipStream_t s, const int* lens, int hop) {
  const dim3 grid((T + 255) / 256, B);
  const dim3 grid16((T + 511) / 512, B);  // 16 outputs per group of 8 lanes
  if (peak) VFX_HIP(hipMemsetAsync(peak, 0, sizeof(unsigned) * B, s));
  switch (C * 2 + (x_f16 ? 1 : 0)) {
    case 64: hipLaunchKernelGGL((k_voc_final<4, false>), grid, dim3(256), 0, s, x, T, w, bias, slope, wav, peak, lens, hop); break;
    case 65: hipLaunchKernelGGL((k_voc_final<4, true>), grid, dim3(256), 0, s, x, T, w, bias, slope, wav, peak, lens, hop); break;
    case 128: hipLaunchKernelGGL((k_voc_final<8, false>), grid, dim3(256), 0, s, x, T, w, bias, slope, wav, peak, lens, hop); break;
    // (32 outputs per group: 1.85 ms -- the 38-row loop no longer stays in registers)
    case 129: hipLaunchKernelGGL((k_voc_final<8, true, 16>), grid16, dim3(256), 0, s, x, T, w, bias, slope, wav, peak, lens, hop); break;
    case 256: hipLaunchKernelGGL((k_voc_final<16, false>), grid, dim3(256), 0, s, x, T, w, bias, slope, wav, peak, lens, hop); break;
    case 257: hipLaunchKernelGGL((k_voc_final<16, true>), grid, dim3(256), 0, s, x, T, w, bias, slope, wav, peak, lens, hop); break;
    default: VFX_CHECK(false, "vocoder tail: %d channels are not supported (32, 64 or 128)", C);
  }
  VFX_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// handler() glue
// ---------------------------------------------------------------------------------------------
// from_log (pytorch_util.py:161-163): 10 ** min(x, 5).  With `sums` != null also accumulates the
// per-clip energy of mel bins 5..24 of the estimate and of the input (amp_to_original_f,
// tools/utils.py:50-55): sums[2b] += est, sums[2b+1] += target.
// lens_t != nullptr (batches of clips of unequal length): only the lens_t[b] frames clip b really has count for its energies
__global__ void k_from_log(const float* __restrict__ logmel, const float* __restrict__ mel_in, int T, int64_t total,
                           float* __restrict__ sums, float* __restrict__ mel_out, const int* __restrict__ lens_t) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float e = 0.f, g = 0.f;
  int64_t b = 0;
  if (idx < total) {
    const float v = exp10f(fminf(logmel[idx], 5.f));
    mel_out[idx] = v;
    const int f = idx & 127;
    b = (idx >> 7) / T;
    const int t = (int)((idx >> 7) - b * T);
    if (sums && f >= 5 && f < 25 && (!lens_t || t < lens_t[b])) {
      e = v;
      g = mel_in[idx];
    }
  }
  if (sums) {
    // a block may straddle two clips only at a clip boundary; keep it simple: per-lane atomics
    // are restricted to the 20 contributing bins.
    if (e != 0.f || g != 0.f) {
      atomicAdd(sums + 2 * b, e);
      atomicAdd(sums + 2 * b + 1, g);
    }
  }
}

__global__ void k_scale_by_ratio(float* __restrict__ mel, int T, int64_t total, const float* __restrict__ sums) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int64_t b = (idx >> 7) / T;
  mel[idx] *= sums[2 * b + 1] / sums[2 * b];
}

void launch_from_log(const float* logmel, const float* mel_in, int B, int T, int unify, float* sums, float* mel_out,
                     hipStream_t s, const int* lens_t) {
  const int64_t total = (int64_t)B * T * 128;
  if (unify) VFX_HIP(hipMemsetAsync(sums, 0, sizeof(float) * 2 * B, s));
  hipLaunchKernelGGL(k_from_log, dim3(nblocks(total, 256)), dim3(256), 0, s, logmel, mel_in, T, total,
                     unify ? sums : nullptr, mel_out, lens_t);
  if (unify) hipLaunchKernelGGL(k_scale_by_ratio, dim3(nblocks(total, 256)), dim3(256), 0, s, mel_out, T, total, sums);
  VFX_HIP(hipGetLastError());
}

// Peak normalise (eval_gsr_voicefixer.py:68-70) + trim_center (tools/utils.py:57-70), per clip.
__global__ void k_absmax(const float* __restrict__ x, int64_t n_per, unsigned* __restrict__ peak) {
  const int b = blockIdx.y;
  const float* xb = x + (int64_t)b * n_per;
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_per; i += (int64_t)gridDim.x * 256)
    m = fmaxf(m, fabsf(xb[i]));
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(peak + b, __float_as_uint(m));
}

__global__ void k_trim_scale(const float* __restrict__ x, int64_t Llong, int L, int off, const unsigned* __restrict__ peak,
                             float* __restrict__ out, int* __restrict__ flags) {
  const int b = blockIdx.y;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= L) return;
  const float p = __uint_as_float(peak[b]);
  // eval_gsr_voicefixer.py:68-70 prints "Warning: Exceed energy limit" here: the handlers read this sticky bit once per file
  if (n == 0 && p > 1.0f && flags) or_flag_global(flags, VFX_FLAG_PEAK_NORMALISED);
  const float v = x[(int64_t)b * Llong + off + n];
  out[(int64_t)b * L + n] = p > 1.0f ? v / p : v;
}

// The same for a batch of clips of unequal length: clip b was restored from lens_l[b] samples through lens_tp[b] vocoder frames,
// so its vocoder output has lens_tp[b] * hop samples and trim_center takes its own centre; the rest of its row of L is zero.
__global__ void k_trim_scale_varlen(const float* __restrict__ x, int64_t Llong, int L, int hop, const int* __restrict__ lens_l,
                                    const int* __restrict__ lens_tp, const unsigned* __restrict__ peak,
                                    float* __restrict__ out, int* __restrict__ flags) {
  const int b = blockIdx.y;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= L) return;
  const float p = __uint_as_float(peak[b]);
  if (n == 0 && p > 1.0f && flags) or_flag_global(flags, VFX_FLAG_PEAK_NORMALISED);
  const int Lb = lens_l[b];
  const int off = (int)(((int64_t)lens_tp[b] * hop - Lb) / 2);
  float v = 0.f;
  if (n < Lb) {
    v = x[(int64_t)b * Llong + off + n];
    v = p > 1.0f ? v / p : v;
  }
  out[(int64_t)b * L + n] = v;
}

// The per-clip lengths of a varlen call travel as KERNEL ARGUMENTS (64 clips per launch): in stream order behind the previous
// call's kernels, no host synchronisation, no pageable-memory copy (and legal inside a stream capture).
struct LensChunk {
  int v[3][64];
};
__global__ void k_set_lens(int* __restrict__ d_lens, int cap, int first, int n, LensChunk c) {
  const int i = threadIdx.x;
  if (i < n)
#pragma unroll
    for (int r = 0; r < 3; ++r) d_lens[r * cap + first + i] = c.v[r][i];
}
void launch_set_lens(int* d_lens, int cap, const int* host /*[3][B]*/, int B, hipStream_t s) {
  for (int first = 0; first < B; first += 64) {
    LensChunk c{};
    const int n = std::min(64, B - first);
    for (int r = 0; r < 3; ++r)
      for (int i = 0; i < n; ++i) c.v[r][i] = host[(size_t)r * B + first + i];
    hipLaunchKernelGGL(k_set_lens, dim3(1), dim3(64), 0, s, d_lens, cap, first, n, c);
  }
  VFX_HIP(hipGetLastError());
}

// rows (b, t < lens_t[b]) of src (B, T, F) -> dst; the other rows of dst are zero (the log-mel estimate of a varlen batch)
__global__ void k_copy_rows_masked(const float* __restrict__ src, float* __restrict__ dst, int T, int F, int64_t total,
                                   const int* __restrict__ lens_t) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int64_t r = idx / F;
  const int64_t b = r / T;
  const int t = (int)(r - b * T);
  dst[idx] = t < lens_t[b] ? src[idx] : 0.f;
}
void launch_copy_rows_masked(const float* src, float* dst, int B, int T, int F, const int* lens_t, hipStream_t s) {
  const int64_t total = (int64_t)B * T * F;
  hipLaunchKernelGGL(k_copy_rows_masked, dim3(nblocks(total, 256)), dim3(256), 0, s, src, dst, T, F, total, lens_t);
  VFX_HIP(hipGetLastError());
}

// A ResUNet group of a varlen call (api.cpp: vfx_restore_gsr_varlen): the rows of the group's clips idx[j] of src (B, T, F) as a
// compact (n, Tg, F) tensor (rows past T: zeros), and back (rows < min(T, Tg)).
__global__ void k_gather_rows(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ dst, int T, int Tg, int F4,
                              int64_t total4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int64_t r = i / F4;
  const int f = (int)(i - r * F4);
  const int j = (int)(r / Tg), t = (int)(r - (int64_t)j * Tg);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (t < T) v = reinterpret_cast<const f32x4*>(src)[((int64_t)idx[j] * T + t) * F4 + f];
  reinterpret_cast<f32x4*>(dst)[i] = v;
}
__global__ void k_scatter_rows(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ dst, int T, int Tg, int F4,
                               int64_t total4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int64_t r = i / F4;
  const int f = (int)(i - r * F4);
  const int j = (int)(r / Tg), t = (int)(r - (int64_t)j * Tg);
  if (t < T) reinterpret_cast<f32x4*>(dst)[((int64_t)idx[j] * T + t) * F4 + f] = reinterpret_cast<const f32x4*>(src)[i];
}
void launch_gather_rows(const float* src, const int* idx, float* dst, int n, int T, int Tg, int F, hipStream_t s) {
  const int64_t total4 = (int64_t)n * Tg * (F / 4);
  hipLaunchKernelGGL(k_gather_rows, dim3(nblocks(total4, 256)), dim3(256), 0, s, src, idx, dst, T, Tg, F / 4, total4);
  VFX_HIP(hipGetLastError());
}
void launch_scatter_rows(const float* src, const int* idx, float* dst, int n, int T, int Tg, int F, hipStream_t s) {
  const int64_t total4 = (int64_t)n * Tg * (F / 4);
  hipLaunchKernelGGL(k_scatter_rows, dim3(nblocks(total4, 256)), dim3(256), 0, s, src, idx, dst, T, Tg, F / 4, total4);
  VFX_HIP(hipGetLastError());
}

void launch_peak_trim_varlen(const float* wav_long, int B, int64_t Llong, int L, int hop, const int* lens_l, const int* lens_tp,
                             const float* peak, float* out, hipStream_t s, int* flags) {
  hipLaunchKernelGGL(k_trim_scale_varlen, dim3((L + 255) / 256, B), dim3(256), 0, s, wav_long, Llong, L, hop, lens_l, lens_tp,
                     reinterpret_cast<const unsigned*>(peak), out, flags);
  VFX_HIP(hipGetLastError());
}

__global__ void k_or_flags(int* flags, int bits) { or_flag_global(flags, bits); }  // (no FLAT atomics anywhere: conv_common.h)
void launch_or_flags(int* flags, int bits, hipStream_t s) {
  if (!flags) return;
  hipLaunchKernelGGL(k_or_flags, dim3(1), dim3(1), 0, s, flags, bits);
  VFX_HIP(hipGetLastError());
}

// `have_peak`: ws[b] already holds the peak of clip b (written by the vocoder tail).
void launch_peak_trim(const float* wav_long, int B, int64_t Llong, int L, float* ws, bool have_peak, float* out,
                      hipStream_t s, int* flags) {
  if (!have_peak) {
    VFX_HIP(hipMemsetAsync(ws, 0, sizeof(unsigned) * B, s));
    const int gx = (int)std::min<int64_t>(64, (Llong + 255) / 256);
    hipLaunchKernelGGL(k_absmax, dim3(gx, B), dim3(256), 0, s, wav_long, Llong, reinterpret_cast<unsigned*>(ws));
  }
  const int64_t diff = Llong - L;
  const int off = (int)(diff / 2);
  hipLaunchKernelGGL(k_trim_scale, dim3((L + 255) / 256, B), dim3(256), 0, s, wav_long, Llong, L, off,
                     reinterpret_cast<const unsigned*>(ws), out, flags);
  VFX_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------
// long-audio chunkers (tools/dsp/overlapadd.py, tools/dsp/overlapadd_boxcar.py)
// ---------------------------------------------------------------------------------------------
// F.unfold with zero padding (overlapadd.py:421-428; overlapadd_boxcar.py:436-452):
//   chunks[b][k][i] = x[b][k * hop - lead + i], zero outside [0, L).
// One thread per 4 consecutive chunk samples; 16-byte loads when the chunk grid is 16-byte aligned.
template <bool VEC>
__global__ void k_chunk_gather(const float* __restrict__ x, int L, int win, int hop, int lead, int n_chunks,
                               float* __restrict__ chunks) {
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= win) return;
  const int k = blockIdx.y, b = blockIdx.z;
  const int64_t src = (int64_t)k * hop - lead + i;
  const float* xb = x + (int64_t)b * L;
  float* dst = chunks + ((int64_t)b * n_chunks + k) * win + i;
  if (VEC) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (src >= 0 && src + 4 <= L) v = *reinterpret_cast<const f32x4*>(xb + src);
    *reinterpret_cast<f32x4*>(dst) = v;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (i + e < win) dst[e] = (src + e >= 0 && src + e < L) ? xb[src + e] : 0.f;
  }
}

void launch_chunk_gather(const float* x, int B, int L, int win, int hop, int lead, int n_chunks, float* chunks,
                         hipStream_t s) {
  const dim3 grid(nblocks(win, 1024), n_chunks, B);
  const bool vec = (win % 4 == 0) && (hop % 4 == 0) && (lead % 4 == 0) && (L % 4 == 0) &&
                   (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (reinterpret_cast<uintptr_t>(chunks) % 16 == 0);
  if (vec)
    hipLaunchKernelGGL(k_chunk_gather<true>, grid, dim3(256), 0, s, x, L, win, hop, lead, n_chunks, chunks);
  else
    hipLaunchKernelGGL(k_chunk_gather<false>, grid, dim3(256), 0, s, x, L, win, hop, lead, n_chunks, chunks);
  VFX_HIP(hipGetLastError());
}

// Synthesis window (or 1 / (win / hop) scale) + F.fold (overlapadd.py:455-471; overlapadd_boxcar.py:494-507):
//   y[b][n] = sum_k frames[b][k][n + lead - k * hop] * w[n + lead - k * hop]   over the chunks that cover n,
// summed in ascending k (a gather, so no atomics and a fixed order).
__global__ void k_chunk_ola(const float* __restrict__ frames, const float* __restrict__ window, float scale,
                            int n_chunks, int win, int hop, int lead, int L, float* __restrict__ y) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= L) return;
  const int b = blockIdx.y;
  const int pos = n + lead;
  const int k1 = min(n_chunks - 1, pos / hop);
  const int k0 = max(0, (pos - win + hop) / hop);  // ceil((pos - win + 1) / hop) for pos - win + 1 > 0
  const float* fb = frames + (int64_t)b * n_chunks * win;
  float acc = 0.f;
  for (int k = k0; k <= k1; ++k) {
    const int i = pos - k * hop;
    if (i < 0 || i >= win) continue;
    const float f = fb[(int64_t)k * win + i];
    acc += window ? f * window[i] : f * scale;
  }
  y[(int64_t)b * L + n] = acc;
}

void launch_chunk_ola(const float* frames, const float* window, float scale, int B, int n_chunks, int win, int hop,
                      int lead, int L, float* y, hipStream_t s) {
  hipLaunchKernelGGL(k_chunk_ola, dim3(nblocks(L, 256), B), dim3(256), 0, s, frames, window, scale, n_chunks, win,
                     hop, lead, L, y);
  VFX_HIP(hipGetLastError());
}

// Debug aid (VFX_DEBUG_NAN): number of non-finite floats in p[0..n).  Synchronises the stream.
__global__ void k_count_nonfinite(const float* __restrict__ p, int64_t n, unsigned long long* __restrict__ count) {
  unsigned long long c = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const unsigned u = __float_as_uint(p[i]);
    c += ((u >> 23) & 0xff) == 0xff;
  }
  if (c) atomicAdd(count, c);
}

int64_t count_nonfinite(const float* p, int64_t n, hipStream_t s) {
  unsigned long long* d = nullptr;
  VFX_HIP(hipMalloc(&d, sizeof(*d)));
  VFX_HIP(hipMemsetAsync(d, 0, sizeof(*d), s));
  hipLaunchKernelGGL(k_count_nonfinite, dim3((unsigned)std::min<int64_t>(4096, (n + 255) / 256)), dim3(256), 0, s, p, n, d);
  unsigned long long hcount = 0;
  VFX_HIP(hipMemcpyAsync(&hcount, d, sizeof(hcount), hipMemcpyDeviceToHost, s));
  VFX_HIP(hipStreamSynchronize(s));
  VFX_HIP(hipFree(d));
  return (int64_t)hcount;
}

// ---------------------------------------------------------------------------------------------
// Spectral metrics of the evaluation handlers (evaluation_proc/metrics.py:83-95, utils.py:81-101), per clip:
//   LSD    = mean_t sqrt( mean_f log10( tgt^2 / (est + 1e-12)^2 + 1e-12 )^2 )
//   SiSpec = 10 log10( |s tgt|^2 / (|est - s tgt|^2 + 1e-12) + 1e-12 ),  s = <est, tgt> / (|tgt|^2 + 1e-8)
// One pass over the data: a wave per frame forms the frame's LSD term and the three inner products in double
// precision (float x float is exact in double, so |est - s tgt|^2 = <e,e> - 2 s <e,t> + s^2 <t,t> does not cancel
// away at 60 dB); the per-clip reduction runs in a fixed order.  HBM-bound: 8 bytes in per bin.
// ---------------------------------------------------------------------------------------------
__global__ void k_metric_frames(const float* __restrict__ est, const float* __restrict__ tgt, int T, int F,
                                double* __restrict__ ws /*[B][T][4]*/) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (t >= T) return;
  const float* e = est + ((int64_t)b * T + t) * F;
  const float* g = tgt + ((int64_t)b * T + t) * F;
  double ee = 0.0, eg = 0.0, gg = 0.0, ls = 0.0;
  for (int f = lane; f < F; f += 64) {
    const float ev = e[f], gv = g[f];
    ee += (double)ev * (double)ev;
    eg += (double)ev * (double)gv;
    gg += (double)gv * (double)gv;
    const float d = ev + 1e-12f;
    const float l = log10f(gv * gv / (d * d) + 1e-12f);
    ls += (double)(l * l);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ee += __shfl_xor(ee, o);
    eg += __shfl_xor(eg, o);
    gg += __shfl_xor(gg, o);
    ls += __shfl_xor(ls, o);
  }
  if (lane == 0) {
    double* w = ws + ((int64_t)b * T + t) * 4;
    w[0] = ee;
    w[1] = eg;
    w[2] = gg;
    w[3] = sqrt(ls / (double)F);
  }
}

__global__ void k_metric_final(const double* __restrict__ ws, int T, float* __restrict__ out /*[B][2]*/) {
  __shared__ double red[4][256];
  const int b = blockIdx.x, tid = threadIdx.x;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int t = tid; t < T; t += 256) {
    const double* w = ws + ((int64_t)b * T + t) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] += w[k];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) red[k][tid] = acc[k];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o)
#pragma unroll
      for (int k = 0; k < 4; ++k) red[k][tid] += red[k][tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    const double ee = red[0][0], eg = red[1][0], gg = red[2][0];
    const double s = eg / (gg + 1e-8);
    const double tt = s * s * gg;
    double nn = ee - 2.0 * s * eg + tt;
    nn = nn < 0.0 ? 0.0 : nn;
    out[2 * b] = (float)(red[3][0] / (double)T);
    out[2 * b + 1] = (float)(10.0 * log10(tt / (nn + 1e-12) + 1e-12));
  }
}

void launch_spectral_metrics(const float* est, const float* tgt, int B, int T, int F, double* ws, float* out, hipStream_t s) {
  hipLaunchKernelGGL(k_metric_frames, dim3((T + 3) / 4, B), dim3(256), 0, s, est, tgt, T, F, ws);
  hipLaunchKernelGGL(k_metric_final, dim3(B), dim3(256), 0, s, ws, T, out);
  VFX_HIP(hipGetLastError());
}

}  // namespace vfx
