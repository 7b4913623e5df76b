// stft.hip -- fused STFT front-end / ISTFT back-end for gfx950.
//
// Forward (vfx_stft_mel): ONE WAVE per frame.  A wave walks F consecutive frames of one clip (F chosen so that the launch is
//   about one round of resident waves); four waves share a workgroup's tables -- the FFT twiddles, the periodic Hann window
//   and the non-zeros of the mel filterbank in LDS -- and nothing else: after the table fill there is no block barrier.
//   Per frame: reflect-padded frame x window  ->  1024-point complex FFT by the wave (64 lanes x 16 points: Stockham passes of
//   radix 16, 16, 4, three exchanges through the wave's own 8.7 KB of LDS)  ->  real-FFT untangle to 1025 bins  ->
//   mag = sqrt(max(re^2+im^2, eps)), cos = re/mag, sin = im/mag (FDomainHelper.spectrogram_phase,
//   tools/pytorch/modules/fDomainHelper.py:60-65)  ->  banded sparse mel projection from LDS (MelScale.forward,
//   tools/pytorch/mel_scale.py:52-64; the filterbank has 2018 non-zeros, 1..55 per band; a lane sums band `lane` and band
//   `127 - lane`)  ->  optional log10(max(.,1e-8)) (to_log, tools/pytorch/pytorch_util.py:157-159).  The samples of frame
//   t+1 are requested before the FFT of frame t.
//   The reference does the DFT as two conv1d(1->1025, k=2048) = 4.2 MMAC/frame; the FFT path needs 79 kflop/frame.  With
//   2276 B/frame of algorithmic traffic (441 new samples in, 128 mel out) the mel-only front-end is bound by VALU ISSUE, not
//   by HBM (35 flop/B against a ridge of 20): a wave64 instruction occupies its SIMD for four cycles and a frame is ~2300 of
//   them.  Measured (16 x 10 s): 65 us = 0.07 of the HBM roofline / 0.12 of the fp32 vector peak; the phase-emitting form
//   (14 KB/frame, HBM-bound) 59 us = 0.48 of 8 TB/s.  History: one frame per workgroup with everything re-fetched 121 us;
//   a 256-thread workgroup walking F frames with a five-pass radix-4 FFT (two block barriers per pass) 90 / 71 us --
//   7 us of barrier and LDS latency per frame and only five workgroups per CU to hide it.
//
// Inverse (vfx_istft): ONE kernel.  A workgroup owns IH = 16 hops of the overlap-add buffer (LDS) and runs the frames that
//   reach into them (IH + 4 = 20; IH = 2 for launches too small to fill the chip; the 4 halo frames are recomputed by the
//   neighbouring workgroup -- their spectra come from L2), one frame per wave at a time: Hermitian spectrum -> packed
//   1024-point complex inverse FFT (the same wave-level transform) -> x synthesis window -> added into the buffer.  Frames
//   that run together are at least five apart, so they touch disjoint samples, and a sample's contributions arrive in a fixed
//   order: five rounds, one block barrier each.  At the end every owned sample is divided by the window sum-of-squares
//   envelope (summed from the window on the fly) and written once.  The 16 KB-per-frame buffer of windowed frames that a
//   two-kernel form moves through HBM (write + read) does not exist.
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

// ---- wave-level 1024-point complex FFT -----------------------------------------------------------------------------------
// ONE wave transforms a frame: 64 lanes x 16 points, Stockham passes of radix 16, 16, 4 (the 16-point butterfly = two
// radix-4 stages in registers), the three exchanges through the wave's own LDS buffer -- no block barrier anywhere: a
// wave's LDS operations execute in order, so a `wave_sync()` (compiler fence) between a store and the loads of other lanes'
// data is all it takes.  The 256-thread form (five radix-4 passes, two block barriers each, one frame per workgroup at a
// time) kept the chip at a handful of frames in flight per CU: 7 us per frame of barrier and LDS latency for 0.3 us of
// arithmetic.
constexpr int ZP = NC + NC / 16 + 1;  // float2 per wave buffer: one pad element per 16 (conflict-free strided access) + 1 (below)
__device__ __forceinline__ int zpad(int i) { return i + (i >> 4); }
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// 16-point DFT in registers, n = 4a + b -> k = c + 4d: radix-4 over a, twiddle W16^(bc), radix-4 over b.
// On return X[k] sits in v[4 (k & 3) + (k >> 2)] (see out16()).
template <int SIGN>
__device__ __forceinline__ void fft16(float2 (&v)[16]) {
#pragma unroll
  for (int b = 0; b < 4; ++b) fft4<SIGN>(v[b], v[4 + b], v[8 + b], v[12 + b]);  // v[4c + b] = Y_b[c]
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
  constexpr float s = SIGN < 0 ? -1.f : 1.f;  // W16^m = cos(2 pi m / 16) + s i sin(2 pi m / 16)
  auto tw = [&](float2& x, float c, float sn) __attribute__((always_inline)) { x = cmul(x, make_float2(c, s * sn)); };
  tw(v[4 * 1 + 1], C1, S1);    // m = 1
  tw(v[4 * 1 + 2], R2, R2);    // m = 2
  tw(v[4 * 1 + 3], S1, C1);    // m = 3
  tw(v[4 * 2 + 1], R2, R2);    // m = 2
  v[4 * 2 + 2] = SIGN < 0 ? make_float2(v[10].y, -v[10].x) : make_float2(-v[10].y, v[10].x);  // m = 4: -+ i
  tw(v[4 * 2 + 3], -R2, R2);   // m = 6
  tw(v[4 * 3 + 1], S1, C1);    // m = 3
  tw(v[4 * 3 + 2], -R2, R2);   // m = 6
  tw(v[4 * 3 + 3], -C1, -S1);  // m = 9
#pragma unroll
  for (int c = 0; c < 4; ++c) fft4<SIGN>(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);  // v[4c + d] = X[c + 4d]
}
__device__ __forceinline__ constexpr int out16(int k) { return 4 * (k & 3) + (k >> 2); }

// In: v[r] = x[lane + 64 r].  Out: v[m] = X[lane + 64 m] (natural order).  `zw` = this wave's LDS buffer (ZP float2), `twl` =
// the workgroup's LDS copy of e^{-2 pi i m / 1024} (conjugated here for the inverse).
template <int SIGN>
__device__ __forceinline__ void wfft1024(float2 (&v)[16], float2* zw, const float2* twl, int lane) {
  auto twid = [&](int idx) __attribute__((always_inline)) {
    float2 w = twl[idx];
    if (SIGN > 0) w.y = -w.y;
    return w;
  };
  // (padded positions written out: i + (i >> 4) is linear in r for every access pattern below, so each is one base address
  // plus an immediate offset)
  // pass 1: radix 16, Ns = 1 (no twiddles): out[16 lane + r] -> 17 lane + r
  fft16<SIGN>(v);
  float2* const w1 = zw + 17 * lane;
#pragma unroll
  for (int r = 0; r < 16; ++r) w1[r] = v[out16(r)];
  wave_sync();
  // pass 2: radix 16, Ns = 16: in[lane + 64 r] * W256^(k r), k = lane mod 16;  out[16 (lane - k) + k + 16 r]
  const int k = lane & 15;
  const float2* const r2 = zw + lane + (lane >> 4);  // in[lane + 64 r] -> + 68 r
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = r2[68 * r];
#pragma unroll
  for (int r = 1; r < 16; ++r) v[r] = cmul(v[r], twid(4 * k * r));
  wave_sync();
  fft16<SIGN>(v);
  float2* const w2 = zw + 17 * (lane - k) + k;  // out[16 (lane - k) + k + 16 r] -> + 17 r
#pragma unroll
  for (int r = 0; r < 16; ++r) w2[17 * r] = v[out16(r)];
  wave_sync();
  // pass 3: radix 4, Ns = 256: four butterflies per lane, j = lane + 64 q: in[j + 256 r] * W1024^(j r) -> X[j + 256 r]
  float2 u[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int j = lane + 64 * q;
#pragma unroll
    for (int r = 0; r < 4; ++r) u[4 * q + r] = r2[68 * q + 272 * r];  // in[j + 256 r]
#pragma unroll
    for (int r = 1; r < 4; ++r) u[4 * q + r] = cmul(u[4 * q + r], twid(j * r));
    fft4<SIGN>(u[4 * q], u[4 * q + 1], u[4 * q + 2], u[4 * q + 3]);
  }
  wave_sync();  // every lane is done reading zw: the caller may overwrite it
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) v[q + 4 * r] = u[4 * q + r];  // X[lane + 64 (q + 4 r)]
}

// base[idx] with a wave-uniform base and a 32-bit BYTE offset: the scalar-base addressing mode (one VGPR per address)
__device__ __forceinline__ float ldg32(const float* base, unsigned idx) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + 4u * idx);
}

__device__ __forceinline__ int reflect_index(int i, int L) {
  i = i < 0 ? -i : i;
  return i >= L ? 2 * (L - 1) - i : i;
}

// The sixteen sample pairs (x[2n], x[2n+1]), n = lane + 64 r, of the reflect-padded frame t.  `x` is wave-uniform (a scalar
// base): every load is base + one 32-bit lane offset.
__device__ __forceinline__ void load_frame(const float* __restrict__ x, int L, int t, int hop, int lane, float2 (&out)[16]) {
  const int base = t * hop - NFFT / 2;
  if (base >= 0 && base + NFFT <= L) {  // wave-uniform: an interior frame
    const unsigned o = (unsigned)(base + 2 * lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) out[r] = make_float2(ldg32(x, o + 128u * r), ldg32(x, o + 128u * r + 1u));
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int g = base + 2 * (lane + 64 * r);
      out[r] = make_float2(ldg32(x, (unsigned)reflect_index(g, L)), ldg32(x, (unsigned)reflect_index(g + 1, L)));
    }
  }
}

constexpr int STFT_WAVES = 4;  // waves per workgroup: each walks its own frames

// e^{-2 pi i (lane + 64 m) / 2048} = e^{-2 pi i lane / 2048} * e^{-2 pi i m / 32}: the lane's own factor (one table read per
// wave) times a compile-time constant -- no table read per bin
__device__ __forceinline__ float2 rtw_of(float2 lane_w, int m) {
  constexpr float C[16] = {1.f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f,
                           0.55557023301960218f, 0.38268343236508977f, 0.19509032201612825f, 0.f, -0.19509032201612825f,
                           -0.38268343236508977f, -0.55557023301960218f, -0.70710678118654752f, -0.83146961230254524f,
                           -0.92387953251128674f, -0.98078528040323043f};
  constexpr float S[16] = {0.f, 0.19509032201612825f, 0.38268343236508977f, 0.55557023301960218f, 0.70710678118654752f,
                           0.83146961230254524f, 0.92387953251128674f, 0.98078528040323043f, 1.f, 0.98078528040323043f,
                           0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f, 0.55557023301960218f,
                           0.38268343236508977f, 0.19509032201612825f};
  return cmul(lane_w, make_float2(C[m], -S[m]));
}

// SPEC: the launch writes sp / cos / sin (any of them); false = the mel-only front-end of the restore path
template <bool SPEC>
__global__ __launch_bounds__(STFT_WAVES * 64, 2) void k_stft_mel(const float* __restrict__ wav, int L, int T,
                                                                 const float* __restrict__ window,
                                                                 const float2* __restrict__ tw,
                                                                 const float2* __restrict__ rtw,
                                                                 const float* __restrict__ fb_val,
                                                                 const int* __restrict__ fb_start,
                                                                 const int* __restrict__ fb_off, float* __restrict__ mel,
                                                                 float* __restrict__ sp, float* __restrict__ cosp,
                                                                 float* __restrict__ sinp, int log10_mel, int hop, float eps,
                                                                 int F, int groups, int total_groups, int fb_lds,
                                                                 const int* __restrict__ lens) {
  __shared__ float2 twl[NC];
  __shared__ float2 wl[NC];  // the window, as pairs (w[2n], w[2n+1])
  __shared__ float fbl[kMelNnzMax];  // the non-zeros of the mel filterbank (band-major): the band sums read them once per frame
  __shared__ float2 zbuf[STFT_WAVES][ZP];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lane = tid & 63;
#pragma unroll
  for (int r = 0; r < NC / (STFT_WAVES * 64); ++r) {
    twl[tid + r * STFT_WAVES * 64] = tw[tid + r * STFT_WAVES * 64];
    wl[tid + r * STFT_WAVES * 64] = *reinterpret_cast<const float2*>(window + 2 * (tid + r * STFT_WAVES * 64));
  }
  if (mel && fb_lds)  // (a caller-supplied filterbank with more non-zeros than the table holds is read from global memory)
    for (int i = tid; i < fb_off[NMEL]; i += STFT_WAVES * 64) fbl[i] = fb_val[i];
  __syncthreads();  // the only block barrier: from here on the waves are independent
  const int gw = blockIdx.x * STFT_WAVES + wave;  // this wave's group of F consecutive frames of one clip
  if (gw >= total_groups) return;
  const int b = gw / groups, g = gw - b * groups;
  // `lens` (batches of clips of unequal length, vfx_restore_gsr_varlen): clip b holds lens[b] <= L samples in its row of L --
  // its frames, and the reflection at its end, are those of a clip of that length; the rows of the outputs past its
  // last frame are written as zeros (finite input for whatever reads the padded batch)
  const int Lrow = L;
  int Tc = T;  // frames of this clip (T stays the row stride of the outputs)
  if (lens) {
    L = lens[__builtin_amdgcn_readfirstlane(b)];
    Tc = min(T, L / hop + 1);
    const int t0 = g * F, t1 = min(T, t0 + F);
    for (int t = max(t0, Tc); t < t1; ++t) {
      const int64_t row = (int64_t)b * T + t;
      if (mel) {
        mel[row * NMEL + lane] = log10_mel ? -8.f : 0.f;
        mel[row * NMEL + 64 + lane] = log10_mel ? -8.f : 0.f;
      }
      if constexpr (SPEC)
        for (int k = lane; k < NBINS; k += 64) {
          if (sp) sp[row * NBINS + k] = 0.f;
          if (cosp) cosp[row * NBINS + k] = 0.f;
          if (sinp) sinp[row * NBINS + k] = 0.f;
        }
    }
    if (t0 >= Tc) return;
  }
  const int t_begin = g * F, t_end = min(Tc, t_begin + F);
  const float* x = wav + (int64_t)b * Lrow;
  float2* zw = zbuf[wave];
  float* ms = reinterpret_cast<float*>(zw);  // the magnitudes of the frame, over the FFT buffer once it is consumed

  // constants of the frame walk: this lane's two mel bands
  int f0[2], o0[2], nb[2];
  // (band lane and band 127 - lane: the bands grow with the frequency -- 1 .. 55 bins -- so every lane sums about 32 products)
  const int band[2] = {lane, NMEL - 1 - lane};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    f0[h] = fb_start[band[h]];
    o0[h] = fb_off[band[h]];
    nb[h] = fb_off[band[h] + 1] - o0[h];
  }

  const float2 rtw_lane = rtw[lane];
  float2 xin[16];
  load_frame(x, L, t_begin, hop, lane, xin);
  for (int t = t_begin; t < t_end; ++t) {
    asm volatile("" : "+v"(lane));  // the frame-independent LDS addresses are recomputed per frame, not kept (and spilled)
    float2 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float2 ww = wl[lane + 64 * r];
      v[r] = make_float2(xin[r].x * ww.x, xin[r].y * ww.y);
    }
    if (t + 1 < t_end) load_frame(x, L, t + 1, hop, lane, xin);  // in flight during this frame's FFT
    wfft1024<-1>(v, zw, twl, lane);
    // natural order to LDS: the untangle pairs Z[k] with Z[1024 - k], which another lane holds
    float2* const nat = zw + lane + (lane >> 4);  // Z[lane + 64 m] -> + 68 m
#pragma unroll
    for (int m = 0; m < 16; ++m) nat[68 * m] = v[m];
    wave_sync();
    // Z[1024 - (lane + 64 m)] = Z[(64 - lane) + 64 (15 - m)]; lane 0: Z[64 (16 - m)], and Z[1024] = Z[0] is its own v[0]
    // (its read of m = 0 lands on the spare element behind the buffer)
    const int jp = 64 - lane;
    const float2* const par = lane == 0 ? zw + 68 : zw + jp + (jp >> 4);
    float2 zn[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) zn[m] = par[68 * (15 - m)];
    if (lane == 0) zn[0] = v[0];
    wave_sync();  // zw is free: the magnitudes go over it
    __builtin_amdgcn_sched_barrier(0);

    // real-FFT untangle: X[k] = E[k] + W^k O[k], E = (Z[k] + conj Z[N-k]) / 2, O = (Z[k] - conj Z[N-k]) / (2i)
    const int64_t row = ((int64_t)b * T + t) * NBINS;
    auto bin = [&](int k, float2 zk, float2 znk, float2 wk) __attribute__((always_inline)) {
      const float2 e = make_float2(0.5f * (zk.x + znk.x), 0.5f * (zk.y - znk.y));
      const float2 o = make_float2(0.5f * (zk.y + znk.y), -0.5f * (zk.x - znk.x));
      const float2 wo = cmul(wk, o);
      const float re = e.x + wo.x, im = e.y + wo.y;
      // clamp on the POWER (fDomainHelper.py:62); v_sqrt_f32 / v_rcp_f32 (1 ulp) instead of the IEEE expansions
      const float mag = __builtin_amdgcn_sqrtf(fmaxf(re * re + im * im, eps));
      ms[k] = mag;
      if constexpr (!SPEC) return;
      if (sp) sp[row + k] = mag;
      if (cosp || sinp) {
        const float inv = __builtin_amdgcn_rcpf(mag);  // mag = 0 (eps = 0, silent bin): inf, 0 * inf = NaN like 0 / 0
        if (cosp) cosp[row + k] = re * inv;
        if (sinp) sinp[row + k] = im * inv;
      }
    };
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      bin(lane + 64 * m, v[m], zn[m], rtw_of(rtw_lane, m));
      if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // four bins at a time: their temporaries are not 16-fold live
    }
    if (lane == 0) bin(NC, v[0], zn[0], make_float2(-1.f, 0.f));  // Z[1024] = Z[0], W^1024 = -1
    wave_sync();
    if (mel) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // four chains (the loads of the loop do not depend on the sums)
        const float* mp = ms + f0[h];
        auto band_sum = [&](auto fp) __attribute__((always_inline)) {
          float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
          int i = 0;
          for (; i + 3 < nb[h]; i += 4) {
            a0 = fmaf(mp[i], fp[i], a0);
            a1 = fmaf(mp[i + 1], fp[i + 1], a1);
            a2 = fmaf(mp[i + 2], fp[i + 2], a2);
            a3 = fmaf(mp[i + 3], fp[i + 3], a3);
          }
          for (; i < nb[h]; ++i) a0 = fmaf(mp[i], fp[i], a0);
          return (a0 + a2) + (a1 + a3);
        };
        const float acc = fb_lds ? band_sum(fbl + o0[h]) : band_sum(fb_val + o0[h]);
        mel[((int64_t)b * T + t) * NMEL + band[h]] = log10_mel ? log10f(fmaxf(acc, 1e-8f)) : acc;
      }
    }
    wave_sync();  // ms is consumed before the next frame's first pass writes the buffer
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
// A workgroup owns IH hops of the overlap-add buffer (LDS) and runs the frames that reach into them, four at a time: each
// wave packs and inverse-transforms its own frame (wave-level FFT, no barrier) and adds it into the buffer; which frames run
// together is chosen so that they cannot touch the same sample (see the kernel).
// IH = 16 (20 frames per workgroup) for large launches, 2 (6 frames, three times the inverse FFTs but a fraction of the serial
// chain) when there are too few frames to fill the chip (streaming chunks).
__global__ __launch_bounds__(STFT_WAVES * 64, 2) void k_istft(const float* __restrict__ re, const float* __restrict__ im,
                                                              const float* __restrict__ window, const float2* __restrict__ tw,
                                                              const float2* __restrict__ rtw, int T, int L, int hop, int groups,
                                                              int IH, float* __restrict__ wav, const int* __restrict__ lens) {
  __shared__ float2 twl[NC];
  __shared__ float2 wl[NC];  // the synthesis window, as pairs
  __shared__ float2 zbuf[STFT_WAVES][ZP];
  extern __shared__ __attribute__((aligned(16))) float ola[];  // [IH * hop]
  const int b = blockIdx.x / groups, g = blockIdx.x - b * groups;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int lane = tid & 63;
  const int span = IH * hop;
  const int p0 = g * span;                      // first owned position of the un-trimmed overlap-add buffer
  // `lens` (a batch of clips of unequal length): clip b has lens[b] <= L samples, i.e. Tc = lens[b] / hop + 1 <= T frames -- the
  // later frames of its rows of re / im do not exist for it and its samples past lens[b] are written as zeros; T and L stay the
  // row strides
  const int Ts = T, Ls = L;
  if (lens) {
    L = lens[b];
    T = min(T, L / hop + 1);
  }
  const int t_lo = max(0, g * IH - (NFFT - 1) / hop), t_hi = min(T - 1, g * IH + IH - 1);
  for (int i = tid; i < span; i += STFT_WAVES * 64) ola[i] = 0.f;
#pragma unroll
  for (int r = 0; r < NC / (STFT_WAVES * 64); ++r) {
    twl[tid + r * STFT_WAVES * 64] = tw[tid + r * STFT_WAVES * 64];
    wl[tid + r * STFT_WAVES * 64] = *reinterpret_cast<const float2*>(window + 2 * (tid + r * STFT_WAVES * 64));
  }
  const float sc = 1.0f / NC;
  const float2 rtw_lane = rtw[lane];
  __syncthreads();  // ola zeroed, twl and wl complete

  // the spectrum of this wave's frame of a round: (Re, Im) of bins k and 1024 - k, k = lane + 64 r
  float2 xk[16], xn[16];
  auto load_spectrum = [&](int t) __attribute__((always_inline)) {
    const float* R = re + ((int64_t)b * Ts + t) * NBINS;
    const float* I = im + ((int64_t)b * Ts + t) * NBINS;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned k = (unsigned)(lane + 64 * r);
      xk[r] = make_float2(ldg32(R, k), ldg32(I, k));
      xn[r] = make_float2(ldg32(R, NC - k), ldg32(I, NC - k));
    }
  };
  // Frames at least D apart touch disjoint samples.  Round r (D rounds, one block barrier each) takes the frames t = r (mod D)
  // -- the ABSOLUTE frame index -- of the workgroup's n frames, wave w every fourth of them: the frames of a round add into the
  // buffer CONCURRENTLY and never meet, and a sample's contributions arrive in the order of t mod D -- sums that depend neither
  // on timing nor on how the buffer is cut into groups (IH, which the launch picks from the batch size): a clip's samples are
  // bit-identical whatever batch it is restored in.
  const int D = (NFFT + hop - 1) / hop;
  const int n = t_hi - t_lo + 1;
  // this wave's frame list in processing order: (r, f) with f = ((r - t_lo) mod D) + D (wave + 4 j), f = t - t_lo
  auto first_in_round = [&](int r) __attribute__((always_inline)) { return ((r - t_lo) % D + D) % D + D * wave; };
  int r = 0, f = first_in_round(0);
  while (r < D && f >= n) f = first_in_round(++r);  // the first frame of this wave, if any
  if (r < D) load_spectrum(t_lo + f);
  for (int round = 0; round < D; ++round) {  // block-uniform trip count
    while (r == round) {
      const int t = t_lo + f;
      asm volatile("" : "+v"(lane));  // as in the forward kernel
      float2 v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float2 e = make_float2(0.5f * (xk[q].x + xn[q].x), 0.5f * (xk[q].y - xn[q].y));
        const float2 d = make_float2(0.5f * (xk[q].x - xn[q].x), 0.5f * (xk[q].y + xn[q].y));
        float2 wk = rtw_of(rtw_lane, q);
        wk.y = -wk.y;  // conj(W^k) = e^{+2 pi i k / 2048}
        const float2 o = cmul(wk, d);
        v[q] = make_float2(e.x - o.y, e.y + o.x);  // Z = E + i O
      }
      // advance to this wave's next frame and request its spectrum: it travels during this frame's FFT
      f += D * STFT_WAVES;
      while (r < D && f >= n) f = first_in_round(++r);
      if (r < D) load_spectrum(t_lo + f);
      wfft1024<1>(v, zbuf[wave], twl, lane);
      const int off = t * hop - p0;  // frame sample o lands at owned index off + o
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const int i0 = off + 2 * (lane + 64 * m);
        const float2 ww = wl[lane + 64 * m];
        if (i0 >= 0 && i0 < span) ola[i0] += v[m].x * sc * ww.x;  // one lane per sample, no other frame of the round nearby
        if (i0 + 1 >= 0 && i0 + 1 < span) ola[i0 + 1] += v[m].y * sc * ww.y;
      }
    }
    __syncthreads();
  }
  const int total = NFFT + hop * (T - 1);
  for (int i = tid; i < span; i += STFT_WAVES * 64) {
    const int p = p0 + i, n = p - NFFT / 2;
    if (n < 0 || n >= Ls) continue;
    float out = 0.f;
    if (p < total && n < L) {
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
    wav[(int64_t)b * Ls + n] = out;
  }
}

// Frames per wave of the forward kernel: as many as keep the launch at about one round of resident waves
// (2 workgroups of 4 waves per CU -- 59 KB of LDS -- x 256 CUs), at most 32.
static int frames_per_group(int64_t frames) {
  const int64_t f = (frames + 2047) / 2048;
  return (int)std::max<int64_t>(1, std::min<int64_t>(32, f));
}

void launch_stft_mel(const FrontEndTables& t, const float* wav, int B, int L, int T, float* mel, float* sp,
                     float* cosp, float* sinp, int log10_mel, int hop, float eps, hipStream_t stream, const int* lens) {
  const int F = frames_per_group((int64_t)B * T);
  const int groups = (T + F - 1) / F;
  const int total = B * groups;
  auto kernel = (sp || cosp || sinp) ? k_stft_mel<true> : k_stft_mel<false>;
  hipLaunchKernelGGL(kernel, dim3((total + STFT_WAVES - 1) / STFT_WAVES), dim3(STFT_WAVES * 64), 0, stream, wav, L, T, t.window,
                     reinterpret_cast<const float2*>(t.twiddle), reinterpret_cast<const float2*>(t.rtwiddle),
                     t.fb_val, t.fb_start, t.fb_off, mel, sp, cosp, sinp, log10_mel, hop, eps, F, groups, total,
                     t.fb_nnz <= kMelNnzMax ? 1 : 0, lens);
  VFX_HIP(hipGetLastError());
}

void launch_mel_project(const FrontEndTables& t, const float* sp, int64_t rows, float* mel, hipStream_t stream) {
  hipLaunchKernelGGL(k_mel_project, dim3((unsigned)rows), dim3(128), 0, stream, sp, rows, t.fb_val, t.fb_start,
                     t.fb_off, mel);
  VFX_HIP(hipGetLastError());
}

void launch_istft(const FrontEndTables& t, const float* re, const float* im, int B, int T, int L, int hop, float* wav,
                  hipStream_t stream, const int* lens) {
  // groups cover the positions [0, 1024 + L) of the un-trimmed overlap-add buffer
  const int IH = (int64_t)B * T >= 4096 ? 16 : 2;
  const int span = IH * hop;
  const int groups = (NFFT / 2 + L + span - 1) / span;
  hipLaunchKernelGGL(k_istft, dim3(B * groups), dim3(256), (size_t)span * sizeof(float), stream, re, im, t.window,
                     reinterpret_cast<const float2*>(t.twiddle), reinterpret_cast<const float2*>(t.rtwiddle), T, L, hop,
                     groups, IH, wav, lens);
  VFX_HIP(hipGetLastError());
}

}  // namespace vfx
