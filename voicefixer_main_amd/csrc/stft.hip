// stft.hip -- fused STFT front-end / ISTFT back-end for gfx950.
//
// Forward (vfx_stft_mel): a 256-thread workgroup walks F consecutive frames of one clip (F chosen so that the whole
//   launch is about one round of resident workgroups).  What does not depend on the frame is fetched ONCE per
//   workgroup -- the periodic Hann window (8 values per thread, registers), the twiddle table of the five radix-4
//   passes (LDS), the band bounds of the mel filterbank -- and the samples of frame t+1 are
//   requested before the FFT of frame t starts, so a frame costs LDS traffic and barriers only:
//   reflect-padded frame x window  ->  1024-point complex radix-4 Stockham FFT in LDS (5 passes, one butterfly per
//   thread per pass)  ->  real-FFT untangle to 1025 bins  ->  mag = sqrt(max(re^2+im^2, eps)), cos = re/mag,
//   sin = im/mag (FDomainHelper.spectrogram_phase, tools/pytorch/modules/fDomainHelper.py:60-65)  ->  banded sparse
//   mel projection from LDS (MelScale.forward, tools/pytorch/mel_scale.py:52-64; the filterbank has 2018 non-zeros,
//   1..55 per band)  ->  optional log10(max(.,1e-8)) (to_log, tools/pytorch/pytorch_util.py:157-159).
//   The reference does the DFT as two conv1d(1->1025, k=2048) = 4.2 MMAC/frame; the FFT needs ~0.06 MFLOP/frame,
//   which makes the stage HBM/latency bound: 441 new samples in, 128 mel out per frame (2276 B/frame algorithmic)
//   when sp/cos/sin are not requested.  (One frame per workgroup, everything re-fetched per frame: 0.30 TB/s.)
//
// Inverse (vfx_istft): ONE kernel.  A workgroup owns IH = 8 hops of the overlap-add buffer and runs the frames that
//   reach into them (IH + floor(2047 / hop) = 12 frames; IH = 2 for launches too small to fill the chip; the 4 halo frames are recomputed by the neighbouring
//   workgroup -- their spectra come from L2): Hermitian spectrum -> packed 1024-point complex inverse FFT -> x synthesis
//   window -> added into the workgroup's slice of the overlap-add buffer IN LDS; at the end every owned sample is
//   divided by the window sum-of-squares envelope (summed from the window on the fly) and written once.  The
//   16 KB-per-frame buffer of windowed frames that a two-kernel form moves through HBM (write + read) does not exist.
//   torchlibrosa ISTFT semantics (see oracle/dsp.py): `y[:, n_fft//2 : n_fft//2 + length]`.
#include "vfx_internal.h"

namespace vfx {

constexpr int NFFT = 2048;
constexpr int NC = NFFT / 2;     // complex FFT length
constexpr int NBINS = NC + 1;    // 1025
constexpr int NMEL = 128;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// In-place (register) radix-4 butterfly.  SIGN = -1: forward (e^{-i}), +1: inverse.
template <int SIGN>
__device__ __forceinline__ void fft4(float2& v0, float2& v1, float2& v2, float2& v3) {
  const float2 a0 = make_float2(v0.x + v2.x, v0.y + v2.y);
  const float2 a1 = make_float2(v0.x - v2.x, v0.y - v2.y);
  const float2 a2 = make_float2(v1.x + v3.x, v1.y + v3.y);
  const float2 d = make_float2(v1.x - v3.x, v1.y - v3.y);
  // forward: d * (-i) = (d.y, -d.x);  inverse: d * (+i) = (-d.y, d.x)
  const float2 a3 = SIGN < 0 ? make_float2(d.y, -d.x) : make_float2(-d.y, d.x);
  v0 = make_float2(a0.x + a2.x, a0.y + a2.y);
  v1 = make_float2(a1.x + a3.x, a1.y + a3.y);
  v2 = make_float2(a0.x - a2.x, a0.y - a2.y);
  v3 = make_float2(a1.x - a3.x, a1.y - a3.y);
}

// 1024-point Stockham FFT, 256 threads, data in `z` (LDS, natural order in and out).
// `v` holds this thread's four inputs of the FIRST pass (z[j + r*256]); `twl` = the workgroup's LDS copy of
// e^{-2 pi i m / 1024} (pass p multiplies input r by twl[k r 1024 / (4 Ns)], Ns = 4^p, k = j mod Ns; conjugated for the
// inverse).  The caller guarantees that nobody reads `z` any more when this is entered; all of `z` is valid (and a
// barrier has been passed) on return.
template <int SIGN>
__device__ __forceinline__ void fft1024(float2* z, float2 v[4], const float2* twl, int j) {
#pragma unroll
  for (int pass = 0; pass < 5; ++pass) {
    const int Ns = 1 << (2 * pass);
    const int k = j & (Ns - 1);
    if (pass > 0) {
      const int tstep = NC / (4 * Ns);  // table stride of this pass
      float2 w[3];
#pragma unroll
      for (int r = 1; r < 4; ++r) {
        w[r - 1] = twl[k * r * tstep];
        if (SIGN > 0) w[r - 1].y = -w[r - 1].y;
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = z[j + r * (NC / 4)];
#pragma unroll
      for (int r = 1; r < 4; ++r) v[r] = cmul(v[r], w[r - 1]);
      __syncthreads();
    }
    fft4<SIGN>(v[0], v[1], v[2], v[3]);
    const int j0 = ((j - k) << 2) + k;
#pragma unroll
    for (int r = 0; r < 4; ++r) z[j0 + r * Ns] = v[r];
  }
  __syncthreads();
}

__device__ __forceinline__ int reflect_index(int i, int L) {
  i = i < 0 ? -i : i;
  return i >= L ? 2 * (L - 1) - i : i;
}

// The four sample pairs (x[2n], x[2n+1]), n = j + 256 r, of the reflect-padded frame t.
__device__ __forceinline__ void load_frame(const float* __restrict__ x, int L, int t, int hop, int j, float2 out[4]) {
  const int base = t * hop - NFFT / 2;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int g = base + 2 * (j + r * (NC / 4));
    if (g >= 0 && g + 1 < L) out[r] = make_float2(x[g], x[g + 1]);
    else out[r] = make_float2(x[reflect_index(g, L)], x[reflect_index(g + 1, L)]);
  }
}

__global__ __launch_bounds__(256, 5) void k_stft_mel(const float* __restrict__ wav, int L, int T,
                                                   const float* __restrict__ window,
                                                   const float2* __restrict__ tw,
                                                   const float2* __restrict__ rtw,
                                                   const float* __restrict__ fb_val,
                                                   const int* __restrict__ fb_start,
                                                   const int* __restrict__ fb_off, float* __restrict__ mel,
                                                   float* __restrict__ sp, float* __restrict__ cosp,
                                                   float* __restrict__ sinp, int log10_mel, int hop, float eps,
                                                   int F, int groups) {
  __shared__ float2 z[NC];
  __shared__ float2 twl[NC];
  __shared__ float mag_s[NBINS + 3];
  const int b = blockIdx.x / groups, g = blockIdx.x - b * groups;
  const int t_begin = g * F, t_end = min(T, t_begin + F);
  const int j = threadIdx.x;
  const float* x = wav + (int64_t)b * L;

  // constants of the whole frame walk: the FFT twiddle table in LDS, the window in registers
  float2 w[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    twl[j + r * (NC / 4)] = tw[j + r * (NC / 4)];
    w[r] = *reinterpret_cast<const float2*>(window + 2 * (j + r * (NC / 4)));
  }
  int f0 = 0, o0 = 0, nb = 0;
  if (j < NMEL) {
    f0 = fb_start[j];
    o0 = fb_off[j];
    nb = fb_off[j + 1] - o0;
  }

  float2 xin[4];
  if (t_begin < t_end) load_frame(x, L, t_begin, hop, j, xin);
  __syncthreads();  // twl is complete (fft1024 reads it before its first barrier)
  for (int t = t_begin; t < t_end; ++t) {
    float2 v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = make_float2(xin[r].x * w[r].x, xin[r].y * w[r].y);
    if (t + 1 < t_end) load_frame(x, L, t + 1, hop, j, xin);  // in flight during this frame's FFT
    fft1024<-1>(z, v, twl, j);

    // real-FFT untangle: X[k] = E[k] + W^k O[k], E = (Z[k] + conj Z[N-k]) / 2, O = (Z[k] - conj Z[N-k]) / (2i)
    const int64_t row = ((int64_t)b * T + t) * NBINS;
    auto bin = [&](int k, float2 wk) __attribute__((always_inline)) {
      const float2 zk = z[k & (NC - 1)];
      const float2 zn = z[(NC - k) & (NC - 1)];
      const float2 e = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
      const float2 o = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
      const float2 wo = cmul(wk, o);
      const float re = e.x + wo.x, im = e.y + wo.y;
      // clamp on the POWER (fDomainHelper.py:62).  v_sqrt_f32 / v_rcp_f32 (1 ulp) instead of the IEEE expansions: the
      // kernel is bound by instruction issue (~700 per wave and frame), and these were a fifth of them
      const float mag = __builtin_amdgcn_sqrtf(fmaxf(re * re + im * im, eps));
      mag_s[k] = mag;
      if (sp) sp[row + k] = mag;
      if (cosp || sinp) {
        const float inv = __builtin_amdgcn_rcpf(mag);  // mag = 0 (eps = 0, silent bin): inf, 0 * inf = NaN like 0 / 0
        if (cosp) cosp[row + k] = re * inv;
        if (sinp) sinp[row + k] = im * inv;
      }
    };
#pragma unroll
    for (int r = 0; r < 4; ++r) bin(j + r * (NC / 4), rtw[j + r * (NC / 4)]);  // the same four L1 lines every frame
    if (j == 0) bin(NC, rtw[NC]);
    __syncthreads();  // mag_s complete; every read of z is done before the next frame's first pass writes it
    if (mel && j < NMEL) {
      float acc0 = 0.f, acc1 = 0.f;  // two chains: the loads of the loop do not depend on the sums
      int i = 0;
      for (; i + 1 < nb; i += 2) {
        acc0 = fmaf(mag_s[f0 + i], fb_val[o0 + i], acc0);
        acc1 = fmaf(mag_s[f0 + i + 1], fb_val[o0 + i + 1], acc1);
      }
      if (i < nb) acc0 = fmaf(mag_s[f0 + i], fb_val[o0 + i], acc0);
      const float acc = acc0 + acc1;
      mel[((int64_t)b * T + t) * NMEL + j] = log10_mel ? log10f(fmaxf(acc, 1e-8f)) : acc;
    }
    // the next write of mag_s (next frame's untangle) lies behind the barriers of the next FFT, which the mel threads
    // only reach after this loop body
  }
}

__global__ __launch_bounds__(128) void k_mel_project(const float* __restrict__ sp, int64_t rows,
                                                      const float* __restrict__ fb_val,
                                                      const int* __restrict__ fb_start,
                                                      const int* __restrict__ fb_off, float* __restrict__ mel) {
  __shared__ float s[NBINS + 3];
  const int64_t row = blockIdx.x;
  for (int k = threadIdx.x; k < NBINS; k += 128) s[k] = sp[row * NBINS + k];
  __syncthreads();
  const int j = threadIdx.x;
  const int f0 = fb_start[j], o0 = fb_off[j], n = fb_off[j + 1] - o0;
  float acc = 0.f;
  for (int i = 0; i < n; ++i) acc = fmaf(s[f0 + i], fb_val[o0 + i], acc);
  mel[row * NMEL + j] = acc;
}

// Inverse.  Per frame: x[n] (n < 2048) = irfft(X)[n] * window[n]:
//   pack Z[k] = E[k] + i O[k] with E = (X[k] + conj X[N-k]) / 2, O = conj(W^k) (X[k] - conj X[N-k]) / 2;
//   z = IFFT1024(Z) / 1024;  x[2n] = Re z[n], x[2n+1] = Im z[n].
// Overlap-add: wav[b, n] = (sum_t x_t[p - t*hop]) / (sum_t window[p - t*hop]^2), p = n + 1024, for every n < L inside the
// overlap-add buffer (p < 2048 + hop*(T-1)); 0 beyond it.  This is torchlibrosa's `y[:, n_fft//2 : n_fft//2 + length]`
// (same in the in-repo twin tools/dsp/base.py:193-200, `end = start + length`): the L mod hop samples past hop*(T-1) are
// reconstructed from the tails of the last frames.  Positions whose envelope is tiny are left undivided
// (librosa.filters.window_sumsquare semantics).
// IH = hops of the overlap-add buffer a workgroup owns: 8 (12 frames per workgroup) for large launches, 2 (6 frames, three
// times the inverse FFTs but a third of the serial chain) when there are too few frames to fill the chip (streaming chunks)

__device__ __forceinline__ void load_spectrum(const float* __restrict__ R, const float* __restrict__ I, int j, float2 xk[4],
                                              float2 xn[4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = j + r * (NC / 4);
    xk[r] = make_float2(R[k], I[k]);
    xn[r] = make_float2(R[NC - k], I[NC - k]);
  }
}

__global__ __launch_bounds__(256, 5) void k_istft(const float* __restrict__ re, const float* __restrict__ im,
                                                const float* __restrict__ window, const float2* __restrict__ tw,
                                                const float2* __restrict__ rtw, int T, int L, int hop, int groups,
                                                int IH, float* __restrict__ wav) {
  __shared__ float2 z[NC];
  __shared__ float2 twl[NC];
  extern __shared__ __attribute__((aligned(16))) float ola[];  // [IH * hop]
  const int b = blockIdx.x / groups, g = blockIdx.x - b * groups;
  const int j = threadIdx.x;
  const int span = IH * hop;
  const int p0 = g * span;                      // first owned position of the un-trimmed overlap-add buffer
  const int t_lo = max(0, g * IH - (NFFT - 1) / hop), t_hi = min(T - 1, g * IH + IH - 1);
  for (int i = j; i < span; i += 256) ola[i] = 0.f;

  float2 w[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    twl[j + r * (NC / 4)] = tw[j + r * (NC / 4)];
    w[r] = *reinterpret_cast<const float2*>(window + 2 * (j + r * (NC / 4)));
  }
  const float sc = 1.0f / NC;

  float2 xk[4], xn[4];
  if (t_lo <= t_hi) {
    const int64_t f = (int64_t)b * T + t_lo;
    load_spectrum(re + f * NBINS, im + f * NBINS, j, xk, xn);
  }
  __syncthreads();  // ola zeroed, twl complete
  for (int t = t_lo; t <= t_hi; ++t) {
    float2 v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float2 e = make_float2(0.5f * (xk[r].x + xn[r].x), 0.5f * (xk[r].y - xn[r].y));
      const float2 d = make_float2(0.5f * (xk[r].x - xn[r].x), 0.5f * (xk[r].y + xn[r].y));
      float2 wk = rtw[j + r * (NC / 4)];
      wk.y = -wk.y;  // conj(W^k) = e^{+2 pi i k / 2048}
      const float2 o = cmul(wk, d);
      v[r] = make_float2(e.x - o.y, e.y + o.x);  // Z = E + i O
    }
    if (t < t_hi) {  // the next frame's spectrum travels during this frame's FFT
      const int64_t f = (int64_t)b * T + t + 1;
      load_spectrum(re + f * NBINS, im + f * NBINS, j, xk, xn);
    }
    fft1024<1>(z, v, twl, j);
    const int off = t * hop - p0;  // frame sample o lands at owned index off + o
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = j + r * (NC / 4);
      const float2 zz = z[n];
      const int i0 = off + 2 * n;
      if (i0 >= 0 && i0 < span) ola[i0] += zz.x * sc * w[r].x;          // one thread per sample of the frame,
      if (i0 + 1 >= 0 && i0 + 1 < span) ola[i0 + 1] += zz.y * sc * w[r].y;  // frames one after the other
    }
    __syncthreads();  // the adds are done and z is free before the next frame's first pass
  }
  const int total = NFFT + hop * (T - 1);
  for (int i = j; i < span; i += 256) {
    const int p = p0 + i, n = p - NFFT / 2;
    if (n < 0 || n >= L) continue;
    float out = 0.f;
    if (p < total) {
      float env = 0.f;
      int th = p / hop;                        // last frame starting at or before p
      if (th > T - 1) th = T - 1;
      int tl = (p - NFFT + hop) / hop;         // first frame with t*hop + 2048 > p
      if (tl < 0) tl = 0;
      for (int t = tl; t <= th; ++t) {
        const int o = p - t * hop;
        if (o >= 0 && o < NFFT) {
          const float ww = window[o];
          env = fmaf(ww, ww, env);
        }
      }
      out = env > 1.1754944e-38f ? ola[i] / env : ola[i];
    }
    wav[(int64_t)b * L + n] = out;
  }
}

// Frames per workgroup of the forward kernel: as many as keep the launch at one round of resident workgroups
// (5 per CU -- 92 VGPRs, 20 KB of LDS -- x 256 CUs), at most 32.
static int frames_per_group(int64_t frames) {
  const int64_t f = (frames + 1279) / 1280;
  return (int)std::max<int64_t>(1, std::min<int64_t>(32, f));
}

void launch_stft_mel(const FrontEndTables& t, const float* wav, int B, int L, int T, float* mel, float* sp,
                     float* cosp, float* sinp, int log10_mel, int hop, float eps, hipStream_t stream) {
  const int F = frames_per_group((int64_t)B * T);
  const int groups = (T + F - 1) / F;
  hipLaunchKernelGGL(k_stft_mel, dim3(B * groups), dim3(256), 0, stream, wav, L, T, t.window,
                     reinterpret_cast<const float2*>(t.twiddle), reinterpret_cast<const float2*>(t.rtwiddle),
                     t.fb_val, t.fb_start, t.fb_off, mel, sp, cosp, sinp, log10_mel, hop, eps, F, groups);
  VFX_HIP(hipGetLastError());
}

void launch_mel_project(const FrontEndTables& t, const float* sp, int64_t rows, float* mel, hipStream_t stream) {
  hipLaunchKernelGGL(k_mel_project, dim3((unsigned)rows), dim3(128), 0, stream, sp, rows, t.fb_val, t.fb_start,
                     t.fb_off, mel);
  VFX_HIP(hipGetLastError());
}

void launch_istft(const FrontEndTables& t, const float* re, const float* im, int B, int T, int L, int hop, float* wav,
                  hipStream_t stream) {
  // groups cover the positions [0, 1024 + L) of the un-trimmed overlap-add buffer
  const int IH = (int64_t)B * T >= 4096 ? 8 : 2;
  const int span = IH * hop;
  const int groups = (NFFT / 2 + L + span - 1) / span;
  hipLaunchKernelGGL(k_istft, dim3(B * groups), dim3(256), (size_t)span * sizeof(float), stream, re, im, t.window,
                     reinterpret_cast<const float2*>(t.twiddle), reinterpret_cast<const float2*>(t.rtwiddle), T, L, hop,
                     groups, IH, wav);
  VFX_HIP(hipGetLastError());
}

}  // namespace vfx
