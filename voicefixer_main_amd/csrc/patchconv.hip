// patchconv.hip -- tap convolution with the input patch staged ONCE per 32-channel chunk.
//
// Same GEMM view, prologue / epilogue semantics and arithmetic modes as k_tapconv
// (tapconv.hip), for layers whose taps overlap: 3x3 Conv2d, the parity classes of the stride-2
// ConvTranspose2d, Conv1d k3 with dilation <= 48, Conv1d k7, ConvTranspose1d phases.
//
//   M tile  = TH x TW block of the logical output grid of ONE image (TH * TW = 128;
//             16x8 / 8x16 / 4x32 for the 2-D levels, 128x1 for the 1-D vocoder layers);
//   patch   = the (TH + dh span) x (TW + dw span) input pixels all taps of the tile touch
//             (<= 224 rows of 32 channels).  Per chunk the patch is loaded from global ONCE,
//             run through the prologue (BN affine + activation, zero halo AFTER the activation,
//             reflect addressing) and, in split-bf16 mode, through the hi/lo split ONCE, then
//             every tap reads its shifted window straight from LDS: A-staging VALU work, global
//             load instructions and L2 traffic drop by the tap overlap (~6.4x for 3x3, ~3x for k3).
//   B tiles = one (tap, chunk) weight tile per K step, register-staged two steps ahead as in
//             k_tapconv (pure 16-byte copies: the weights are pre-split on the host).
//   The next chunk's patch is prefetched into registers right after the current one is
//   installed, i.e. it has a whole chunk (ntaps K steps) to land.
#include "conv_epilogue.h"
#include "vfx_internal.h"

namespace vfx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int PBM = 128;
constexpr int PLDK = kKC + 4;          // LDS row length in floats (144 bytes)
constexpr int NQ = kPatchMaxRows / 32;  // patch row groups per thread

#define VFX_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ f32x4 p_ldg4(const float* p) { return *(const VFX_GLOBAL f32x4*)p; }
__device__ __forceinline__ void p_stg4(float* p, f32x4 v) { *(VFX_GLOBAL f32x4*)p = v; }

template <int BN, bool ELU, bool SPLIT>
__global__ __launch_bounds__(256, 2) void k_patchconv(const TapConvParams* __restrict__ pp) {
  constexpr int WAVES_N = BN >= 64 ? 2 : 1;
  constexpr int WAVES_M = 4 / WAVES_N;
  constexpr int WM = PBM / (32 * WAVES_M);
  constexpr int WN = BN / (32 * WAVES_N);
  constexpr int BP = BN / 32;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ap = smem;                                   // [kPatchMaxRows][PLDK]
  float* Bs = smem + kPatchMaxRows * PLDK;            // [2][BN][PLDK]
  int* otab = reinterpret_cast<int*>(Bs + 2 * BN * PLDK);  // [128]

  const TapConvParams& p = *pp;
  const int tid = threadIdx.x;
  const int n_tiles = p.Cout / BN;
  int tile;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int n0 = (tile % n_tiles) * BN;
  int mt = tile / n_tiles;                 // spatial tile: (image, tile row, tile col), col fastest
  const int tj = mt % p.tiles_w;
  mt /= p.tiles_w;
  const int ti = mt % p.tiles_h;
  const int img = mt / p.tiles_h;
  const int i0 = ti * p.TH, j0 = tj * p.TW;
  const int Hi = p.Hi, Wi = p.Wi, PW = p.PW, P = p.P;

  // ---- per-thread roles ---------------------------------------------------------------------------
  const int lr = tid >> 3, cg = tid & 7;
  int pix[NQ];          // source pixel index of patch row lr + 32q (0 when out of range)
  unsigned okmask = 0;  // bit q: row is inside the image
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int prow = lr + 32 * q;
    const int pi = prow / PW, pj = prow - pi * PW;
    const int si = i0 + p.dh_min + pi;
    int sj = j0 + p.dw_min + pj;
    int rj = sj < 0 ? -sj : sj;
    rj = rj >= Wi ? 2 * (Wi - 1) - rj : rj;
    sj = p.reflect_w ? rj : sj;
    const bool ok = (prow < P) & ((unsigned)si < (unsigned)Hi) & ((unsigned)sj < (unsigned)Wi);
    pix[q] = ok ? (img * Hi + si) * Wi + sj : 0;
    okmask |= ok ? (1u << q) : 0u;
  }
  if (tid < PBM) {
    const int li = tid >> p.tw_shift, lj = tid & (p.TW - 1);
    const int i = i0 + li, j = j0 + lj;
    int idx = -1;
    if (i < p.Hg && j < p.Wg) {
      const int oh = i * p.sh + p.oh0, ow = j * p.sw + p.ow0;
      if (oh < p.Ho && ow < p.Wo) idx = (img * p.Ho + oh) * p.Wo + ow;
    }
    otab[tid] = idx;
  }

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
  int arow[WM];  // patch row of this lane's pixel in M block a (tap offset added per step)
#pragma unroll
  for (int a = 0; a < WM; ++a) {
    const int ml = (wm * WM + a) * 32 + l31;
    arow[a] = (ml >> p.tw_shift) * PW + (ml & (p.TW - 1));
  }

  // ---- patch (A) staging ------------------------------------------------------------------------
  f32x4 pa[NQ], psc, psh;
  float pslope = 1.f;
  int ps_seg = 0, ps_chunk = 0;     // chunk whose patch is loaded NEXT
  int patches_left = 0;
  for (int s = 0; s < p.nseg; ++s) patches_left += p.seg[s].C / kKC;

  auto issue_patch = [&]() {
    const TapSeg& S = p.seg[ps_seg];
    const int C = S.C;
    const int c0 = ps_chunk * kKC + 4 * cg;
#pragma unroll
    for (int q = 0; q < NQ; ++q) pa[q] = p_ldg4(S.src + (int64_t)pix[q] * C + c0);
    psc = p_ldg4(S.scale + c0);
    psh = p_ldg4(S.shift + c0);
    pslope = S.act == ACT_NONE ? 1.f : S.slope;
    --patches_left;
    if (patches_left > 0) {
      ++ps_chunk;
      if (ps_chunk * kKC == C) {
        ps_chunk = 0;
        ++ps_seg;
      }
    }
  };

  auto store_patch = [&]() {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = pa[q][e] * psc[e] + psh[e];
        float u;
        if constexpr (ELU) u = t > 0.f ? t : expm1f(t);
        else u = t >= 0.f ? t : t * pslope;
        v[e] = (okmask & (1u << q)) ? u : 0.f;
      }
      float* rowf = Ap + (lr + 32 * q) * PLDK;
      if constexpr (SPLIT) {
        const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2));
        const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2], v[3]}, bf16x2));
        const f32x2 r01 = {v[0] - __builtin_bit_cast(float, h01 << 16), v[1] - __builtin_bit_cast(float, h01 & 0xffff0000u)};
        const f32x2 r23 = {v[2] - __builtin_bit_cast(float, h23 << 16), v[3] - __builtin_bit_cast(float, h23 & 0xffff0000u)};
        const unsigned l01 = __builtin_bit_cast(unsigned, __builtin_convertvector(r01, bf16x2));
        const unsigned l23 = __builtin_bit_cast(unsigned, __builtin_convertvector(r23, bf16x2));
        char* rowp = reinterpret_cast<char*>(rowf);
        *reinterpret_cast<uint2*>(rowp + 8 * cg) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(rowp + 64 + 8 * cg) = make_uint2(l01, l23);
      } else {
        *reinterpret_cast<f32x4*>(rowf + 4 * cg) = v;
      }
    }
  };

  // ---- weight (B) staging: two register stages, two K steps ahead ----------------------------------
  struct BStage {
    f32x4 rb[BP];
  };
  BStage st0, st1;
  int bs_seg = 0, bs_chunk = 0, bs_tap = 0;
  int b_left = p.total_steps;
  auto issue_b = [&](BStage& R) {
    const TapSeg& S = p.seg[bs_seg];
    const float* wb = S.wt + ((int64_t)(bs_chunk * S.ntaps + bs_tap) * p.Cout + n0) * kKC;
#pragma unroll
    for (int q = 0; q < BP; ++q) R.rb[q] = p_ldg4(wb + (lr + 32 * q) * kKC + 4 * cg);
    --b_left;
    if (b_left > 0) {
      ++bs_tap;
      if (bs_tap == S.ntaps) {
        bs_tap = 0;
        ++bs_chunk;
        if (bs_chunk * kKC == S.C) {
          bs_chunk = 0;
          ++bs_seg;
        }
      }
    }
  };
  auto store_b = [&](int buf, const BStage& R) {
    float* Bb = Bs + buf * BN * PLDK;
#pragma unroll
    for (int q = 0; q < BP; ++q) *reinterpret_cast<f32x4*>(Bb + (lr + 32 * q) * PLDK + 4 * cg) = R.rb[q];
  };

  auto compute = [&](int buf, int toff) {
    if constexpr (SPLIT) {
      const char* A0 = reinterpret_cast<const char*>(Ap) + toff * (PLDK * 4) + 16 * lh;
      const char* Bb = reinterpret_cast<const char*>(Bs + buf * BN * PLDK + (wn * WN * 32 + l31) * PLDK) + 16 * lh;
#pragma unroll
      for (int s = 0; s < kKC / 16; ++s) {
        bf16x8 ah[WM], al[WM], bh[WN], bl[WN];
#pragma unroll
        for (int a = 0; a < WM; ++a) {
          const char* ap = A0 + arow[a] * (PLDK * 4) + 32 * s;
          ah[a] = *reinterpret_cast<const bf16x8*>(ap);
          al[a] = *reinterpret_cast<const bf16x8*>(ap + 64);
        }
#pragma unroll
        for (int b = 0; b < WN; ++b) {
          bh[b] = *reinterpret_cast<const bf16x8*>(Bb + b * 32 * PLDK * 4 + 32 * s);
          bl[b] = *reinterpret_cast<const bf16x8*>(Bb + b * 32 * PLDK * 4 + 64 + 32 * s);
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
      const float* A0 = Ap + toff * PLDK + 4 * lh;
      const float* Bb = Bs + buf * BN * PLDK + (wn * WN * 32 + l31) * PLDK + 4 * lh;
#pragma unroll
      for (int k8 = 0; k8 < kKC / 8; ++k8) {
        f32x4 fa[WM], fb[WN];
#pragma unroll
        for (int a = 0; a < WM; ++a) fa[a] = *reinterpret_cast<const f32x4*>(A0 + arow[a] * PLDK + k8 * 8);
#pragma unroll
        for (int b = 0; b < WN; ++b) fb[b] = *reinterpret_cast<const f32x4*>(Bb + b * 32 * PLDK + k8 * 8);
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

  // ---- main loop -----------------------------------------------------------------------------------
  int cs_seg = 0, cs_tap = 0, cs_chunk = 0;   // step being computed
  int c_left = p.total_steps;
  auto half_step = [&](int buf, BStage& R) {
    const bool live = c_left > 0;
    if (live && cs_tap == 0) {
      __syncthreads();   // every wave is done with the previous chunk's patch
      store_patch();
      issue_patch();     // prefetch the following chunk (re-reads the last one at the very end)
    }
    store_b(buf, R);
    __syncthreads();
    issue_b(R);
    if (live) {
      const TapSeg& S = p.seg[cs_seg];
      compute(buf, S.poff[cs_tap]);
      --c_left;
      ++cs_tap;
      if (cs_tap == S.ntaps) {
        cs_tap = 0;
        ++cs_chunk;
        if (cs_chunk * kKC == S.C) {
          cs_chunk = 0;
          ++cs_seg;
        }
      }
    }
  };

  issue_patch();
  issue_b(st0);
  issue_b(st1);
  const int n_iter = (p.total_steps + 1) >> 1;
  for (int it = 0; it < n_iter; ++it) {
    half_step(0, st0);
    half_step(1, st1);
  }

  // ---- epilogue: bias + residual, channels-last 16-byte stores (conv_epilogue.h) -----------------
  conv_epilogue<BN, WM, WN, WAVES_N>(p, smem, otab, acc, n0);
}

static size_t patch_lds_bytes(int BN) {
  const size_t main_bytes = (size_t)(kPatchMaxRows * PLDK + 2 * BN * PLDK) * 4;
  const size_t epi_bytes = (size_t)PBM * (BN + 4) * 4;
  return std::max(main_bytes, epi_bytes) + PBM * 4;
}

template <int BN, bool ELU, bool SPLIT>
static void launch_one(int grid, hipStream_t stream, const TapConvParams* dparams) {
  const size_t lds = patch_lds_bytes(BN);
  static bool attr_set = false;
  if (!attr_set) {
    VFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_patchconv<BN, ELU, SPLIT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_patchconv<BN, ELU, SPLIT>), dim3(grid), dim3(256), lds, stream, dparams);
}

template <bool ELU, bool SPLIT>
static void launch_bn(int BN, int grid, hipStream_t stream, const TapConvParams* dparams) {
  switch (BN) {
    case 128: launch_one<128, ELU, SPLIT>(grid, stream, dparams); break;
    case 64: launch_one<64, ELU, SPLIT>(grid, stream, dparams); break;
    default: launch_one<32, ELU, SPLIT>(grid, stream, dparams); break;
  }
}

void launch_patchconv(const TapConvParams& hp, const TapConvParams* dparams, hipStream_t stream) {
  VFX_CHECK(hp.use_patch && hp.P <= kPatchMaxRows && hp.TH * hp.TW == PBM, "patchconv: bad patch geometry");
  VFX_CHECK(hp.Cout % 32 == 0, "patchconv: Cout=%d is not a multiple of 32", hp.Cout);
  bool elu = false;
  for (int s = 0; s < hp.nseg; ++s) elu = elu || hp.seg[s].act == ACT_ELU;
  if (elu)
    for (int s = 0; s < hp.nseg; ++s)
      VFX_CHECK(hp.seg[s].act == ACT_ELU, "patchconv: ELU cannot be mixed with other prologues in one launch");
  const int BN = hp.Cout % 128 == 0 ? 128 : (hp.Cout % 64 == 0 ? 64 : 32);
  const int64_t grid = (int64_t)hp.B * hp.tiles_h * hp.tiles_w * (hp.Cout / BN);
  VFX_CHECK(grid > 0 && grid < ((int64_t)1 << 31), "patchconv: bad grid");
  if (hp.split) {
    if (elu) launch_bn<true, true>(BN, (int)grid, stream, dparams);
    else launch_bn<false, true>(BN, (int)grid, stream, dparams);
  } else {
    if (elu) launch_bn<true, false>(BN, (int)grid, stream, dparams);
    else launch_bn<false, false>(BN, (int)grid, stream, dparams);
  }
  VFX_HIP(hipGetLastError());
}

}  // namespace vfx
