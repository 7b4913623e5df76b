// tapconv.hip -- generic "tap convolution" as an implicit GEMM on the gfx950 fp32 MFMA.
//
// One kernel serves every GEMM-shaped layer of the hot path:
//   * 3x3 Conv2d of ConvBlockRes (models/components/modules.py:223-271) incl. its BN+LeakyReLU
//     pre-activation (prologue), the identity / 1x1-conv shortcut (extra K segment on the raw
//     input) and the residual add (epilogue);
//   * the channel concat in DecoderBlockRes4B (modules.py:212-220) as two K segments;
//   * the stride-2 ConvTranspose2d of the decoders as 4 output-parity launches;
//   * the TFGAN vocoder's Conv1d k3 (dilated) / k7 (reflect padded) and its ConvTranspose1d
//     upsamplers as `scale` output-phase launches (H = 1).
//
// GEMM view:  out[m, n] = sum_seg sum_tap sum_c  P(src_seg[pix(m) + off(tap), c]) * W_seg[tap][c][n]
//   M = B*Hg*Wg logical output pixels, N = Cout, K = sum ntaps*C.  Activations are channels-last
//   fp32, so a K step (one tap, 32 channels) is a 128-byte row per pixel.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 -- exact fp32 products, fp32 accumulate (an fmaf chain),
//   because bf16/fp16 operands miss the "log-mel L1 <= 1e-3" bar of the reference by 19x / 3x
//   (DESIGN.md §4).  Peak 157.3 TFLOP/s.
//
// Tiling: block = 256 threads = 4 waves, tile BM=128 pixels x BN in {32,64,128} couts, K step 32.
//   LDS: A[2][128][36] + B[2][BN][36] fp32 (row padded 32->36 floats = 144 B so that the
//   ds_read_b128 fragment reads of 16 consecutive rows fall on 16 distinct 16-byte slots of the
//   256-byte bank row: conflict-free).  Register-staged double buffering: the global loads of
//   step s+1 are issued before the MFMAs of step s and written to the other LDS buffer after
//   them; one barrier per step.
//   MFMA operand trick: lane l supplies k = l>>5 of each 32x32x2 step; a lane reads ONE float4
//   at channel offset 4*(l>>5) and feeds its 4 components to 4 consecutive MFMAs, i.e. the 8
//   channels of a k8 group are consumed in the order (0,4),(1,5),(2,6),(3,7) -- any order is
//   valid as long as A and B agree -- so every LDS read is a 16-byte read.
#include "conv_epilogue.h"
#include "vfx_internal.h"

namespace vfx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 128;
constexpr int LDK = kKC + 4;  // padded LDS row length in floats

// Pointers read out of the parameter block are generic to the compiler (-> flat_load, which
// also ticks the LDS counter); every tensor here lives in global memory, so say so.
#define VFX_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ f32x4 ldg4(const float* p) { return *(const VFX_GLOBAL f32x4*)p; }
__device__ __forceinline__ void stg4(float* p, f32x4 v) { *(VFX_GLOBAL f32x4*)p = v; }

template <int BN, bool ELU, bool SPLIT>
__global__ __launch_bounds__(256, 2) void k_tapconv(const TapConvParams* __restrict__ pp) {
  constexpr int WAVES_N = BN >= 64 ? 2 : 1;
  constexpr int WAVES_M = 4 / WAVES_N;
  constexpr int WM = BM / (32 * WAVES_M);  // 32x32 blocks per wave along M
  constexpr int WN = BN / (32 * WAVES_N);  // and along N
  constexpr int BP = BN / 32;              // B-tile load passes per thread

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                        // [2][BM][LDK]
  float* Bs = smem + 2 * BM * LDK;         // [2][BN][LDK]
  int* otab = reinterpret_cast<int*>(Bs + 2 * BN * LDK);  // [BM] output pixel index or -1

  const TapConvParams& p = *pp;
  const int tid = threadIdx.x;
  const int n_tiles = p.Cout / BN;
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch rule, speed only), so
  // give every XCD a contiguous range of tiles -- neighbouring tiles share halo rows and the
  // N-tiles of one M-tile share the whole activation tile through that XCD's L2.
  int tile;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int m0 = (tile / n_tiles) * BM;
  const int n0 = (tile % n_tiles) * BN;

  const int Hi = p.Hi, Wi = p.Wi, Hg = p.Hg, Wg = p.Wg, M = p.M;

  // ---- per-thread staging roles -------------------------------------------------------------
  const int lr = tid >> 3;  // row within a 32-row pass
  const int cg = tid & 7;   // float4 column group: channels 4*cg .. 4*cg+3 of the K chunk
  int a_i[4], a_j[4], a_b[4];  // logical pixel of the 4 A rows this thread stages (a_b < 0: none)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int m = m0 + lr + 32 * q;
    if (m < M) {
      const int t = m / Wg;
      a_j[q] = m - t * Wg;
      a_b[q] = t / Hg;
      a_i[q] = t - a_b[q] * Hg;
    } else {
      a_i[q] = 0; a_j[q] = 0; a_b[q] = -1;
    }
  }
  if (tid < BM) {
    const int m = m0 + tid;
    int idx = -1;
    if (m < M) {
      const int t = m / Wg;
      const int j = m - t * Wg;
      const int b = t / Hg;
      const int i = t - b * Hg;
      const int oh = i * p.sh + p.oh0, ow = j * p.sw + p.ow0;
      if (oh < p.Ho && ow < p.Wo) idx = (b * p.Ho + oh) * p.Wo + ow;
    }
    otab[tid] = idx;
  }

  // ---- accumulators ---------------------------------------------------------------------------
  f32x16 acc[WM][WN];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int b = 0; b < WN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int l31 = lane & 31, lh = lane >> 5;

  // ---- K-loop state (uniform) ------------------------------------------------------------------
  // Two register stages: the loads of step s+2 are issued while step s is computed, so every
  // global load has two full K steps (plus the co-resident block's) to land.
  struct Stage {
    f32x4 ra[4];
    f32x4 rb[BP];
    f32x4 sc, sh;
    unsigned ok;
    float slope;
  };
  Stage st0, st1;
  int s_seg = 0, s_chunk = 0, s_tap = 0;  // next step to ISSUE

  // Branch-free by construction (fixed number of loads per call) so that the compiler can keep
  // counted s_waitcnt vmcnt(N) waits across the loop back-edge: out-of-range rows read pixel 0
  // and are zeroed when staged; scale/shift are always loaded (identity tables when the segment
  // has no affine prologue); past the last K step the last step is re-read with ok = 0, which
  // stages an all-zero A tile.
  int steps_left = p.total_steps;
  auto issue_loads = [&](Stage& R) {
    const TapSeg& S = p.seg[s_seg];
    const int C = S.C;
    const int dh = S.dh[s_tap], dw = S.dw[s_tap];
    const int c0 = s_chunk * kKC + 4 * cg;
    const unsigned live = steps_left > 0 ? ~0u : 0u;
    unsigned ok_bits = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ii = a_i[q] + dh;
      int jj = a_j[q] + dw;
      int rj = jj < 0 ? -jj : jj;
      rj = rj >= Wi ? 2 * (Wi - 1) - rj : rj;
      jj = p.reflect_w ? rj : jj;
      const bool ok = (a_b[q] >= 0) & ((unsigned)ii < (unsigned)Hi) & ((unsigned)jj < (unsigned)Wi);
      const int pix = ok ? (a_b[q] * Hi + ii) * Wi + jj : 0;
      R.ra[q] = ldg4(S.src + (int64_t)pix * C + c0);
      ok_bits |= ok ? (1u << q) : 0u;
    }
    R.ok = ok_bits & live;
    const float* wb = S.wt + ((int64_t)(s_chunk * S.ntaps + s_tap) * p.Cout + n0) * kKC;
#pragma unroll
    for (int q = 0; q < BP; ++q) R.rb[q] = ldg4(wb + (lr + 32 * q) * kKC + 4 * cg);
    R.sc = ldg4(S.scale + c0);
    R.sh = ldg4(S.shift + c0);
    R.slope = S.act == ACT_NONE ? 1.f : S.slope;  // identity == leaky with slope 1
    // advance (seg, chunk, tap), saturating at the last step: tap innermost so that consecutive
    // steps re-touch the same activation lines (shifted by one pixel) while they are in L1/L2.
    --steps_left;
    if (steps_left > 0) {
      ++s_tap;
      if (s_tap == S.ntaps) {
        s_tap = 0;
        ++s_chunk;
        if (s_chunk * kKC == C) {
          s_chunk = 0;
          ++s_seg;
        }
      }
    }
  };

  auto store_lds = [&](int buf, const Stage& R) {
    float* Ab = As + buf * BM * LDK;
    float* Bb = Bs + buf * BN * LDK;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = R.ra[q][e] * R.sc[e] + R.sh[e];
        float u;
        if constexpr (ELU) u = t > 0.f ? t : expm1f(t);  // only the vocoder's condnet instantiates this
        else u = t >= 0.f ? t : t * R.slope;
        v[e] = (R.ok & (1u << q)) ? u : 0.f;
      }
      if constexpr (SPLIT) {
        // v = hi + lo with hi = bf16(v), lo = bf16(v - hi); row layout [32 hi | 32 lo | pad] (144 B)
        const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2));
        const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2], v[3]}, bf16x2));
        const f32x2 r01 = {v[0] - __builtin_bit_cast(float, h01 << 16), v[1] - __builtin_bit_cast(float, h01 & 0xffff0000u)};
        const f32x2 r23 = {v[2] - __builtin_bit_cast(float, h23 << 16), v[3] - __builtin_bit_cast(float, h23 & 0xffff0000u)};
        const unsigned l01 = __builtin_bit_cast(unsigned, __builtin_convertvector(r01, bf16x2));
        const unsigned l23 = __builtin_bit_cast(unsigned, __builtin_convertvector(r23, bf16x2));
        char* rowp = reinterpret_cast<char*>(Ab + (lr + 32 * q) * LDK);
        *reinterpret_cast<uint2*>(rowp + 8 * cg) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(rowp + 64 + 8 * cg) = make_uint2(l01, l23);
      } else {
        *reinterpret_cast<f32x4*>(Ab + (lr + 32 * q) * LDK + 4 * cg) = v;
      }
    }
#pragma unroll
    for (int q = 0; q < BP; ++q)
      *reinterpret_cast<f32x4*>(Bb + (lr + 32 * q) * LDK + 4 * cg) = R.rb[q];
  };

  auto compute = [&](int buf) {
    if constexpr (SPLIT) {
      // 32x32x16 bf16 MFMA: lane l supplies k = 8*(l>>5) .. +7 of each 16-wide K group.
      const char* Ab = reinterpret_cast<const char*>(As + buf * BM * LDK + (wm * WM * 32 + l31) * LDK) + 16 * lh;
      const char* Bb = reinterpret_cast<const char*>(Bs + buf * BN * LDK + (wn * WN * 32 + l31) * LDK) + 16 * lh;
#pragma unroll
      for (int s = 0; s < kKC / 16; ++s) {
        bf16x8 ah[WM], al[WM], bh[WN], bl[WN];
#pragma unroll
        for (int a = 0; a < WM; ++a) {
          ah[a] = *reinterpret_cast<const bf16x8*>(Ab + a * 32 * LDK * 4 + 32 * s);
          al[a] = *reinterpret_cast<const bf16x8*>(Ab + a * 32 * LDK * 4 + 64 + 32 * s);
        }
#pragma unroll
        for (int b = 0; b < WN; ++b) {
          bh[b] = *reinterpret_cast<const bf16x8*>(Bb + b * 32 * LDK * 4 + 32 * s);
          bl[b] = *reinterpret_cast<const bf16x8*>(Bb + b * 32 * LDK * 4 + 64 + 32 * s);
        }
#pragma unroll
        for (int a = 0; a < WM; ++a)
#pragma unroll
          for (int b = 0; b < WN; ++b) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
          }
      }
    } else {
      const float* Ab = As + buf * BM * LDK + (wm * WM * 32 + l31) * LDK + 4 * lh;
      const float* Bb = Bs + buf * BN * LDK + (wn * WN * 32 + l31) * LDK + 4 * lh;
#pragma unroll
      for (int k8 = 0; k8 < kKC / 8; ++k8) {
        f32x4 fa[WM], fb[WN];
#pragma unroll
        for (int a = 0; a < WM; ++a) fa[a] = *reinterpret_cast<const f32x4*>(Ab + a * 32 * LDK + k8 * 8);
#pragma unroll
        for (int b = 0; b < WN; ++b) fb[b] = *reinterpret_cast<const f32x4*>(Bb + b * 32 * LDK + k8 * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int a = 0; a < WM; ++a)
#pragma unroll
            for (int b = 0; b < WN; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][e], fb[b][e], acc[a][b], 0, 0, 0);
      }
    }
  };

  // ---- main loop (unrolled by two so that the register stages are statically named) ----------
  const int n_iter = (p.total_steps + 1) >> 1;
  issue_loads(st0);
  issue_loads(st1);
  for (int it = 0; it < n_iter; ++it) {
    store_lds(0, st0);
    __syncthreads();
    issue_loads(st0);
    compute(0);
    store_lds(1, st1);
    __syncthreads();
    issue_loads(st1);
    compute(1);
  }

  // ---- epilogue: bias + residual, channels-last 16-byte stores (conv_epilogue.h) -----------------
  conv_epilogue<BN, WM, WN, WAVES_N>(p, smem, otab, acc, n0);
}

static size_t tapconv_lds_bytes(int BN) { return (size_t)(2 * BM * LDK + 2 * BN * LDK) * 4 + BM * 4; }

static int pick_bn(int Cout) {
  if (Cout % 128 == 0) return 128;
  if (Cout % 64 == 0) return 64;
  return 32;
}

template <int BN, bool ELU, bool SPLIT>
static void launch_one(int grid, size_t lds, hipStream_t stream, const TapConvParams* dparams) {
  static bool attr_set = false;
  if (!attr_set) {
    VFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tapconv<BN, ELU, SPLIT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_tapconv<BN, ELU, SPLIT>), dim3(grid), dim3(256), lds, stream, dparams);
}

template <bool ELU, bool SPLIT>
static void launch_bn(int BN, int grid, size_t lds, hipStream_t stream, const TapConvParams* dparams) {
  switch (BN) {
    case 128: launch_one<128, ELU, SPLIT>(grid, lds, stream, dparams); break;
    case 64: launch_one<64, ELU, SPLIT>(grid, lds, stream, dparams); break;
    default: launch_one<32, ELU, SPLIT>(grid, lds, stream, dparams); break;
  }
}

void launch_tapconv(const TapConvParams& hp, const TapConvParams* dparams, hipStream_t stream) {
  VFX_CHECK(hp.Cout % 32 == 0, "tapconv: Cout=%d is not a multiple of 32", hp.Cout);
  bool elu = false;
  for (int s = 0; s < hp.nseg; ++s) {
    VFX_CHECK(hp.seg[s].C % kKC == 0, "tapconv: segment %d has C=%d, not a multiple of %d", s, hp.seg[s].C, kKC);
    elu = elu || hp.seg[s].act == ACT_ELU;
  }
  if (elu)
    for (int s = 0; s < hp.nseg; ++s)
      VFX_CHECK(hp.seg[s].act == ACT_ELU, "tapconv: ELU cannot be mixed with other prologues in one launch");
  VFX_CHECK(hp.M > 0 && hp.total_steps > 0, "tapconv: empty problem");
  const int BN = pick_bn(hp.Cout);
  const int m_tiles = (hp.M + BM - 1) / BM;
  const int grid = m_tiles * (hp.Cout / BN);
  const size_t lds = tapconv_lds_bytes(BN);
  if (hp.split) {
    if (elu) launch_bn<true, true>(BN, grid, lds, stream, dparams);
    else launch_bn<false, true>(BN, grid, lds, stream, dparams);
  } else {
    if (elu) launch_bn<true, false>(BN, grid, lds, stream, dparams);
    else launch_bn<false, false>(BN, grid, lds, stream, dparams);
  }
  VFX_HIP(hipGetLastError());
}

double tapconv_flops(const TapConvParams& hp) {
  double k = 0;
  for (int s = 0; s < hp.nseg; ++s) k += (double)hp.seg[s].ntaps * hp.seg[s].C;
  return 2.0 * (double)hp.M * hp.Cout * k;
}

}  // namespace vfx
