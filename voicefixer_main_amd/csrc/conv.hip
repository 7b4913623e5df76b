// conv.hip -- stage-driven tap convolution (implicit GEMM) for gfx950.
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
//   M = output pixels, N = Cout, K = sum ntaps*C.  Activations are channels-last fp32, so a K
//   step (one tap, 32 channels) is a 128-byte row per pixel.
//
// Work decomposition
//   block   = 4 waves, tile = TH x TW (<= 128) pixels of ONE image x BN in {128, 64, 32} couts;
//   wave    = (128 / WAVES_M) pixels x 32 couts, WAVES_N = BN / 32 (so a wave owns ONE 32-cout
//             column block and BN/32 row blocks);
//   stage   = one 32-channel chunk of one source tensor and the taps that read it.  The host
//             flattens every launch into a table of stages (ConvStage, vfx_internal.h).
//
// Operand paths
//   A (activations): per stage the input PATCH (tile + halo of the stage's taps, <= 224 pixels x
//     32 channels) is loaded from global ONCE, run through the prologue (per-channel affine =
//     folded eval-mode BatchNorm, LeakyReLU / ReLU / ELU, zero halo AFTER the activation,
//     reflect addressing) and, in split-bf16 mode, through the hi/lo split ONCE, then written
//     to one of two LDS patch buffers; every tap reads its shifted window of that buffer.  The
//     patch of stage s+1 is fetched into registers at the start of stage s and written to the
//     other buffer at its end: ONE barrier per stage, none per K step.
//   B (weights): never touch LDS.  They are packed on the host in MFMA fragment order
//     ([chunk][tap][cout/32][fragment][lane] -> one coalesced 1 KB load per fragment per wave) and
//     go global -> VGPR -> MFMA.  The loads of a group of <= 3 taps are issued before the
//     patch loads of the stage (vmcnt is in-order: a later weight wait must not drain the patch).
//
// Arithmetic (vfx_config.precision)
//   1: split-bf16 -- every operand is hi + lo (two bf16), products hi*hi + hi*lo + lo*hi on
//      v_mfma_f32_32x32x16_bf16, fp32 accumulate (~2^-16 relative operand error);
//   0: exact fp32 on v_mfma_f32_32x32x2_f32.
//   Plain bf16 / fp16 operands miss the "log-mel L1 <= 1e-3" bar of the reference.
#include "conv_epilogue.h"
#include "vfx_internal.h"

namespace vfx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int CBM = 128;                // pixels per tile
constexpr int CLDK = kKC + 4;           // LDS row length in floats (144 bytes): conflict-free 16-byte row reads
constexpr int CNQ = kPatchMaxRows / 32; // patch row groups per thread

// Pointers read out of the parameter block are generic to the compiler (-> flat_load, which also
// ticks the LDS counter); every tensor here lives in global memory, so say so.
#define VFX_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ f32x4 c_ldg4(const float* p) { return *(const VFX_GLOBAL f32x4*)p; }

template <int BN, bool ELU, bool SPLIT>
__global__ __launch_bounds__(256, 2) void k_conv(const TapConvParams* __restrict__ pp) {
  constexpr int WAVES_N = BN / 32;
  constexpr int WM = BN / 32;  // 32-row blocks per wave (= 4 / WAVES_M)

  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kMainFloats = (2 * kPatchMaxRows * CLDK > CBM * (BN + 4)) ? 2 * kPatchMaxRows * CLDK : CBM * (BN + 4);
  int* otab = reinterpret_cast<int*>(smem + kMainFloats);  // [128] output pixel index or -1

  const TapConvParams& p = *pp;
  // The stage table is read-only for the whole launch: address it in the constant address space so
  // that every descriptor field is a scalar load (a generic pointer would be read with per-lane
  // flat loads and make the whole stage loop look divergent to the compiler).
  typedef const ConvStage __attribute__((address_space(4))) * StageTab;
  const StageTab stages = (StageTab)(uintptr_t)p.stages;
  const int nstages = p.nstages;
  const int tid = threadIdx.x;
  const int n_tiles = p.Cout / BN;
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch rule, speed only), so
  // give every XCD a contiguous range of tiles -- neighbouring tiles share halo rows and the
  // N-tiles of one spatial tile share the whole patch through that XCD's L2.
  int tile;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int n0 = (tile % n_tiles) * BN;
  int mt = tile / n_tiles;  // spatial tile: (image, tile row, tile col), col fastest
  const int tj = mt % p.tiles_w;
  mt /= p.tiles_w;
  const int ti = mt % p.tiles_h;
  const int img = mt / p.tiles_h;
  const int i0 = ti * p.TH, j0 = tj * p.TW;
  const int Hi = p.Hi, Wi = p.Wi, PW = p.PW, P = p.P;
  const int nq = (P + 31) >> 5;  // patch row groups in use (uniform)

  // ---- per-thread roles ---------------------------------------------------------------------------
  // Every stage of a launch stages the same PH x PW window around the tile (patch pixel -> (row, col)
  // through the pij table); only its origin (dh0, dw0) varies between stages.
  const int lr = tid >> 3, cg = tid & 7;
  const int tw_shift = p.tw_shift, TWm1 = p.TW - 1, TH = p.TH;
  int pij[CNQ];  // (patch row << 16 | patch col) of patch pixel lr + 32q; pixels past P never pass the bounds test
#pragma unroll
  for (int q = 0; q < CNQ; ++q) {
    const int prow = lr + 32 * q;
    const int pi = prow / PW, pj = prow - pi * PW;
    pij[q] = prow < P ? ((pi << 16) | pj) : 0x7fff0000;
  }
  if (tid < CBM) {
    const int li = tid >> tw_shift, lj = tid & TWm1;
    const int i = i0 + li, j = j0 + lj;
    int idx = -1;
    if (li < TH && i < p.Hg && j < p.Wg) {
      const int oh = i * p.sh + p.oh0, ow = j * p.sw + p.ow0;
      if (oh < p.Ho && ow < p.Wo) idx = (img * p.Ho + oh) * p.Wo + ow;
    }
    otab[tid] = idx;
  }

  f32x16 acc[WM][1];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][0][r] = 0.f;

  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int l31 = lane & 31, lh = lane >> 5;
  int aoff[WM];  // byte offset of this lane's pixel of M block a inside a patch buffer (tap offset added per step)
#pragma unroll
  for (int a = 0; a < WM; ++a) {
    const int ml = (wm * WM + a) * 32 + l31;
    const int li = ml >> tw_shift;
    aoff[a] = (li < TH ? li * PW + (ml & TWm1) : 0) * (CLDK * 4) + 16 * lh;
  }
  const int nb_off = ((n0 >> 5) + wn) * 1024 + lane * 4;  // this lane's slot in a weight fragment block

  // ---- patch (A) staging ------------------------------------------------------------------------
  // Loads are unconditional (pixels outside the image or the patch read pixel 0 and are zeroed when
  // staged) so that every stage body is straight-line code with a fixed number of loads: the compiler
  // then keeps counted s_waitcnt vmcnt(N) waits, i.e. the prefetches really stay in flight.
  f32x4 pa[CNQ], psc, psh;
  float pslope = 1.f;
  unsigned okmask = 0;

  auto issue_patch = [&](const __attribute__((address_space(4))) ConvStage& S) {
    const float* src = S.src;
    const int C = S.C;
    const int dh0 = i0 + S.dh0, dw0 = j0 + S.dw0;
    okmask = 0;
#pragma unroll
    for (int q = 0; q < CNQ; ++q) {
      const int si = dh0 + (pij[q] >> 16);
      int sj = dw0 + (pij[q] & 0xffff);
      int rj = sj < 0 ? -sj : sj;
      rj = rj >= Wi ? 2 * (Wi - 1) - rj : rj;
      sj = p.reflect_w ? rj : sj;
      const bool ok = ((unsigned)si < (unsigned)Hi) & ((unsigned)sj < (unsigned)Wi);
      const int pix = ok ? (img * Hi + si) * Wi + sj : 0;
      pa[q] = c_ldg4(src + (int64_t)pix * C + 4 * cg);
      okmask |= ok ? (1u << q) : 0u;
    }
    psc = c_ldg4(S.scale + 4 * cg);
    psh = c_ldg4(S.shift + 4 * cg);
    pslope = S.slope;
  };

  auto store_group = [&](float* Ap, int q, f32x4 raw) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float t = raw[e] * psc[e] + psh[e];
      float u;
      if constexpr (ELU) u = t > 0.f ? t : expm1f(t);  // only the vocoder's condnet instantiates this
      else u = fmaxf(t, t * pslope);                    // LeakyReLU, slope in [0, 1]: 1 = identity, 0 = ReLU
      v[e] = (okmask & (1u << q)) ? u : 0.f;
    }
    float* rowf = Ap + (lr + 32 * q) * CLDK;
    if constexpr (SPLIT) {
      // v = hi + lo with hi = bf16(v), lo = bf16(v - hi); row layout [32 hi | 32 lo | pad] (144 B)
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
  };
  // nqs = row groups to stage (uniform); the guard only skips VALU + LDS work, never a global load
  auto store_patch = [&](float* Ap) {
#pragma unroll
    for (int q = 0; q < CNQ; ++q)
      if (q < 4 || q < nq) store_group(Ap, q, pa[q]);
  };

  // ---- weight (B) fragments: global -> VGPR -----------------------------------------------------
  struct BF {
    f32x4 f[4];  // split: (hi, lo) of k 0..15, (hi, lo) of k 16..31; fp32: the four k8 groups
  };
  auto load_b = [&](BF& R, const float* wtap) {
    const float* b = wtap + nb_off;
#pragma unroll
    for (int i = 0; i < 4; ++i) R.f[i] = c_ldg4(b + i * 256);
  };

  auto compute = [&](const BF& R, const float* Ap, int toff) {
    const char* A0 = reinterpret_cast<const char*>(Ap) + toff * (CLDK * 4);
    if constexpr (SPLIT) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, R.f[2 * s]);
        const bf16x8 bl = __builtin_bit_cast(bf16x8, R.f[2 * s + 1]);
        bf16x8 ah[WM], al[WM];
#pragma unroll
        for (int a = 0; a < WM; ++a) {
          const char* ap = A0 + aoff[a] + 32 * s;
          ah[a] = *reinterpret_cast<const bf16x8*>(ap);
          al[a] = *reinterpret_cast<const bf16x8*>(ap + 64);
        }
        // small cross terms first, the dominant hi*hi product last; consecutive MFMAs hit different accumulators
#pragma unroll
        for (int a = 0; a < WM; ++a) acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl, acc[a][0], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < WM; ++a) acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh, acc[a][0], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < WM; ++a) acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh, acc[a][0], 0, 0, 0);
      }
    } else {
      // 32x32x2 fp32 MFMA: lane l supplies k = l>>5; a lane reads ONE float4 at channel 8g + 4*(l>>5)
      // and feeds its components to 4 consecutive MFMAs (K order (0,4),(1,5),(2,6),(3,7) on both operands).
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 fb = R.f[g];
        f32x4 fa[WM];
#pragma unroll
        for (int a = 0; a < WM; ++a) fa[a] = *reinterpret_cast<const f32x4*>(A0 + aoff[a] + 32 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int a = 0; a < WM; ++a)
            acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][e], fb[e], acc[a][0], 0, 0, 0);
      }
    }
  };

  // ---- stage bodies ------------------------------------------------------------------------------
  // One straight-line body per tap count NT.  Weight fragments live in a ring of three register
  // groups (taps t, t+1, t+2); a group is refilled (tap t+3, or tap 0 of the next stage) right after
  // its tap has been computed.  Order of issue at the start of a stage: weights of taps 1, 2 (tap 0
  // was fetched by the previous stage), THEN the next stage's patch -- vmcnt retires in order, so a
  // wait for a weight fragment issued before the patch loads leaves the patch in flight.
  BF R0, R1, R2;
  auto body = [&](auto NTc, const __attribute__((address_space(4))) ConvStage& S,
                  const __attribute__((address_space(4))) ConvStage& N, const float* Acur, float* Anext) {
    constexpr int NT = decltype(NTc)::value;
    const float* wt = S.wt;
    const int64_t ts = S.tap_stride;
    int poff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) poff[t] = S.poff[t];
    const float* nwt = N.wt;
    if constexpr (NT >= 2) load_b(R1, wt + ts);
    if constexpr (NT >= 3) load_b(R2, wt + 2 * ts);
    issue_patch(N);
    if constexpr (NT == 1) load_b(R1, nwt);  // tap 0 of the next stage: a free ring slot, younger than the patch
    if constexpr (NT == 2) load_b(R2, nwt);
    // the ring slot of tap t is t % 3; `next` must end up in R0
    auto slot = [&](int t) -> BF& { return t % 3 == 0 ? R0 : (t % 3 == 1 ? R1 : R2); };
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      compute(slot(t), Acur, poff[t]);
      __builtin_amdgcn_sched_barrier(0);  // keep the fragment reads of tap t+1 below the MFMAs of tap t (register pressure)
      if (t + 3 < NT) load_b(slot(t), wt + (t + 3) * ts);
      else if (t + 3 == NT) load_b(slot(t), nwt);  // slot (NT % 3): moved to R0 below if it is not R0
    }
    store_patch(Anext);
    if constexpr (NT % 3 == 1) R0 = R1;
    if constexpr (NT % 3 == 2) R0 = R2;
  };

  // ---- stage loop ----------------------------------------------------------------------------------
  issue_patch(stages[0]);
  load_b(R0, stages[0].wt);
  store_patch(smem);

  for (int st = 0; st < nstages; ++st) {
    const __attribute__((address_space(4))) ConvStage& S = stages[st];
    const __attribute__((address_space(4))) ConvStage& N = stages[st + 1 < nstages ? st + 1 : st];  // past the end: refetch (never consumed)
    const float* Acur = smem + (st & 1) * (kPatchMaxRows * CLDK);
    float* Anext = smem + ((st + 1) & 1) * (kPatchMaxRows * CLDK);
    const int ntaps = S.ntaps;
    __syncthreads();  // patch `st` is visible; every wave is done with the buffer patch st+1 will overwrite
    switch (ntaps) {
      case 1: body(std::integral_constant<int, 1>{}, S, N, Acur, Anext); break;
      case 2: body(std::integral_constant<int, 2>{}, S, N, Acur, Anext); break;
      case 3: body(std::integral_constant<int, 3>{}, S, N, Acur, Anext); break;
      case 4: body(std::integral_constant<int, 4>{}, S, N, Acur, Anext); break;
      case 7: body(std::integral_constant<int, 7>{}, S, N, Acur, Anext); break;
      default: body(std::integral_constant<int, 9>{}, S, N, Acur, Anext); break;
    }
  }

  // ---- epilogue: bias + residual, channels-last 16-byte stores (conv_epilogue.h) -----------------
  conv_epilogue<BN, WM, 1, WAVES_N>(p, smem, otab, acc, n0);
}

static size_t conv_lds_bytes(int BN) {
  const size_t main_floats = std::max<size_t>((size_t)2 * kPatchMaxRows * CLDK, (size_t)CBM * (BN + 4));
  return main_floats * 4 + CBM * 4;
}

template <int BN, bool ELU, bool SPLIT>
static void launch_one(int grid, hipStream_t stream, const TapConvParams* dparams) {
  const size_t lds = conv_lds_bytes(BN);
  static bool attr_set = false;
  if (!attr_set) {
    VFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv<BN, ELU, SPLIT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL((k_conv<BN, ELU, SPLIT>), dim3(grid), dim3(256), lds, stream, dparams);
}

template <bool ELU, bool SPLIT>
static void launch_bn(int BN, int grid, hipStream_t stream, const TapConvParams* dparams) {
  switch (BN) {
    case 128: launch_one<128, ELU, SPLIT>(grid, stream, dparams); break;
    case 64: launch_one<64, ELU, SPLIT>(grid, stream, dparams); break;
    default: launch_one<32, ELU, SPLIT>(grid, stream, dparams); break;
  }
}

int conv_block_n(int Cout) { return Cout % 128 == 0 ? 128 : (Cout % 64 == 0 ? 64 : 32); }

void launch_conv(const TapConvParams& hp, const TapConvParams* dparams, hipStream_t stream) {
  VFX_CHECK(hp.nstages > 0 && hp.P > 0 && hp.P <= kPatchMaxRows && hp.TH * hp.TW <= CBM, "conv: bad stage geometry");
  VFX_CHECK(hp.Cout % 32 == 0, "conv: Cout=%d is not a multiple of 32", hp.Cout);
  bool elu = false;
  for (int s = 0; s < hp.nseg; ++s) elu = elu || hp.seg[s].act == ACT_ELU;
  if (elu)
    for (int s = 0; s < hp.nseg; ++s)
      VFX_CHECK(hp.seg[s].act == ACT_ELU, "conv: ELU cannot be mixed with other prologues in one launch");
  const int BN = conv_block_n(hp.Cout);
  const int64_t grid = (int64_t)hp.B * hp.tiles_h * hp.tiles_w * (hp.Cout / BN);
  VFX_CHECK(grid > 0 && grid < ((int64_t)1 << 31), "conv: bad grid");
  if (hp.split) {
    if (elu) launch_bn<true, true>(BN, (int)grid, stream, dparams);
    else launch_bn<false, true>(BN, (int)grid, stream, dparams);
  } else {
    if (elu) launch_bn<true, false>(BN, (int)grid, stream, dparams);
    else launch_bn<false, false>(BN, (int)grid, stream, dparams);
  }
  VFX_HIP(hipGetLastError());
}

double conv_flops(const TapConvParams& hp) {
  double k = 0;
  for (int s = 0; s < hp.nseg; ++s) k += (double)hp.seg[s].ntaps * hp.seg[s].C;
  return 2.0 * (double)hp.M * hp.Cout * k;
}

}  // namespace vfx
