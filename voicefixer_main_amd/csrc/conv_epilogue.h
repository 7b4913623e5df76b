// conv_epilogue.h -- epilogue of k_conv.
//
// k_conv multiplies W (MFMA A operand, rows = couts) by the patch (B operand, columns = pixels): a lane's
// 32x32 accumulator block holds ONE pixel (column = lane & 31) and four runs of 4 consecutive couts
// (row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)).  The runs go through LDS as 16-byte writes (re-using the
// staging buffers) so that the residual read and the output writes are 16-byte-per-lane accesses that
// cover whole channels-last rows (a wave instruction = two or more complete 128-byte lines; writing the runs
// straight from the registers -- 32 pixels x 32 bytes per instruction -- measured 20 % slower).  All residual
// loads of a thread are issued first, all results are formed in registers, and only then are the
// stores issued back-to-back: on gfx950 vmcnt also counts stores, so a store placed between two
// waited loads would serialise on the previous store's acknowledgement.
//
// Two outputs, either optional: the raw fp32 tensor, and the ACTIVATED tensor for a consumer
// convolution (TapConvParams::out_act): y -> affine -> LeakyReLU / ELU -> (split-bf16: hi | lo; 16-bit mode: fp16 | 0).  In
// split mode a lane pair (8 consecutive channels) swaps halves through DPP so that the even lane
// stores the 8 hi values and the odd lane the 8 lo values, 16 bytes each.
#pragma once
#include "vfx_internal.h"

namespace vfx {

typedef float ce_f32x4 __attribute__((ext_vector_type(4)));
typedef float ce_f32x2 __attribute__((ext_vector_type(2)));
typedef float ce_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned ce_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 ce_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 ce_f16x2 __attribute__((ext_vector_type(2)));
#define VFX_CE_GLOBAL __attribute__((address_space(1)))

template <int BN, int WM, int WN, int WAVES_N, bool SPLIT, int HALVES = 1, bool RESACT = false>
__device__ __forceinline__ void conv_epilogue(const TapConvParams& p, float* smem, const int* otab,
                                              ce_f32x16 (&acc)[WM][WN], int n0, float* partial = nullptr) {
  // partial != nullptr (split-K): the raw accumulators go to that slice of the workspace, nothing else happens here
  // (k_splitk_reduce adds the slices and applies bias / residual / activation)
  // HALVES = 2: the cout range goes through the staging buffer in two passes (half the LDS: the 16-bit BN = 128 tile
  // then fits three blocks per CU)
  constexpr int BH = BN / HALVES;  // couts per pass
  constexpr int WPH = WAVES_N / HALVES;  // N-waves per pass
  static_assert(WAVES_N % HALVES == 0 && WPH >= 1, "bad epilogue split");
  constexpr int LDO = BH + 4;  // staged row length (floats), keeps 16-byte alignment
  constexpr int V = BH / 4;    // float4 per output row
  constexpr int RPP = 256 / V; // rows per pass
  constexpr int NPASS = 128 / RPP;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int l31 = lane & 31, lh = lane >> 5;
  const int c4 = tid % V, r0 = tid / V;
  const int Cout = p.Cout;
  const bool even = (tid & 1) == 0;
  const bool f16 = p.hionly != 0;
#pragma unroll
  for (int hh = 0; hh < HALVES; ++hh) {
    __syncthreads();  // every wave is done reading the last stage / the previous pass has been consumed
    // C/D layout of the 32x32 MFMA: col = lane & 31 (-> pixel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (-> cout).
    if (HALVES == 1 || wn / WPH == hh) {
#pragma unroll
      for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int row = (wm * WM + a) * 32 + l31;
            *reinterpret_cast<ce_f32x4*>(smem + row * LDO + ((wn % WPH) * WN + b) * 32 + 8 * j + 4 * lh) =
                ce_f32x4{acc[a][b][4 * j], acc[a][b][4 * j + 1], acc[a][b][4 * j + 2], acc[a][b][4 * j + 3]};
          }
    }
    __syncthreads();
    const int ncol = n0 + hh * BH + 4 * c4;
    ce_f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && !partial) bv = *(const VFX_CE_GLOBAL ce_f32x4*)(p.bias + ncol);
    int opix[NPASS];
    ce_f32x4 val[NPASS];
    // a block of a phased launch that covers several phases (k_conv) keeps one pixel table per phase: the slot of this thread's couts
    const int* otab_s = otab + ((p.nphase > 1 && p.cout_phase < BN) ? (hh * BH + 4 * c4) / p.cout_phase * 128 : 0);
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
      opix[q] = otab_s[r0 + q * RPP];
      val[q] = *reinterpret_cast<const ce_f32x4*>(smem + (r0 + q * RPP) * LDO + 4 * c4) + bv;
    }
    if (partial) {
#pragma unroll
      for (int q = 0; q < NPASS; ++q)
        if (opix[q] >= 0) *(VFX_CE_GLOBAL ce_f32x4*)(partial + (int64_t)opix[q] * Cout + ncol) = val[q];
      continue;
    }
    if constexpr (RESACT) {
      // 16-bit mode on the fp16 trunk: the residual comes as the ACTIVATED fp16 form of it (TapConvParams::residual_act, 8 bytes
      // per lane and row); the raw value is recovered as min(v, v / slope).  A compile-time variant: both residual forms in one
      // body spill in the 168-register BN = 128 tile.  The four halves are bit-cast as a whole vector (hipcc 7.2 miscompiles
      // __builtin_bit_cast of a vector ELEMENT: conv_common.h).
      typedef unsigned ce_u32x2 __attribute__((ext_vector_type(2)));
      typedef _Float16 ce_f16x4 __attribute__((ext_vector_type(4)));
      const float inv = p.residual_inv_slope;
      ce_u32x2 rh[NPASS];
#pragma unroll
      for (int q = 0; q < NPASS; ++q)
        rh[q] = *(const VFX_CE_GLOBAL ce_u32x2*)(reinterpret_cast<const _Float16*>(p.residual_act) +
                                                (int64_t)(opix[q] < 0 ? 0 : opix[q]) * Cout + ncol);
#pragma unroll
      for (int q = 0; q < NPASS; ++q) {
        const ce_f16x4 h = __builtin_bit_cast(ce_f16x4, rh[q]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = (float)h[e];
          val[q][e] += __builtin_fminf(v, v * inv);
        }
      }
    } else if (p.residual) {
      ce_f32x4 res[NPASS];
#pragma unroll
      for (int q = 0; q < NPASS; ++q)
        res[q] = *(const VFX_CE_GLOBAL ce_f32x4*)(p.residual + (int64_t)(opix[q] < 0 ? 0 : opix[q]) * Cout + ncol);
#pragma unroll
      for (int q = 0; q < NPASS; ++q) val[q] += res[q];
    }
    if (p.out) {
#pragma unroll
      for (int q = 0; q < NPASS; ++q)
        if (opix[q] >= 0) *(VFX_CE_GLOBAL ce_f32x4*)(p.out + (int64_t)opix[q] * (p.out_cmul ? p.out_cmul : Cout) + ncol) = val[q];
    }
    if (p.out_act) {
      bool f16_sat = false;  // 16-bit mode: an activation left the fp16 range and was clamped (VFX_FLAG_F16_SATURATED)
      ce_f32x4 asc = {1.f, 1.f, 1.f, 1.f}, ash = {0.f, 0.f, 0.f, 0.f};
      if (p.act_scale) asc = *(const VFX_CE_GLOBAL ce_f32x4*)(p.act_scale + ncol);
      if (p.act_shift) ash = *(const VFX_CE_GLOBAL ce_f32x4*)(p.act_shift + ncol);
      const float slope = p.act_slope;
      const bool elu = p.act_elu != 0;
      // split: the pair's 8 channels live in chunk ncol/32; hi block at +0, lo block at +16 floats, 8 channels = 4 floats
      // 16-bit mode: the activated tensor is an fp16 tensor (2 bytes per element, Cout / 2 floats per pixel); the even
      // lane of a pair stores the pair's 8 consecutive channels (16 bytes) at channel ncol
      const int aoff = (SPLIT && f16) ? (ncol >> 1) : SPLIT ? (ncol & ~31) + ((ncol & 31) >> 3) * 4 + (even ? 0 : 16) : ncol;
      const int64_t astride = (SPLIT && f16) ? (Cout >> 1) : Cout;
#pragma unroll
      for (int q = 0; q < NPASS; ++q) {
        ce_f32x4 u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = val[q][e] * asc[e] + ash[e];
          u[e] = elu ? (t > 0.f ? t : expm1f(t)) : fmaxf(t, t * slope);
        }
        ce_f32x4 o;
        if constexpr (SPLIT) {
          unsigned h01, h23, l01, l23;
          if (f16) {  // 16-bit mode: fp16 (saturating) in the hi half, the lo half is never read
            const ce_f32x2 c01 = {__builtin_fminf(__builtin_fmaxf(u[0], -65504.f), 65504.f), __builtin_fminf(__builtin_fmaxf(u[1], -65504.f), 65504.f)};
            const ce_f32x2 c23 = {__builtin_fminf(__builtin_fmaxf(u[2], -65504.f), 65504.f), __builtin_fminf(__builtin_fmaxf(u[3], -65504.f), 65504.f)};
            h01 = __builtin_bit_cast(unsigned, __builtin_convertvector(c01, ce_f16x2));
            h23 = __builtin_bit_cast(unsigned, __builtin_convertvector(c23, ce_f16x2));
            l01 = l23 = 0u;
            // the predicate of conv_common.h's f16_out_of_range (this header is also used without it)
            f16_sat = f16_sat | !(__builtin_fabsf(u[0]) <= 65504.f) | !(__builtin_fabsf(u[1]) <= 65504.f) |
                      !(__builtin_fabsf(u[2]) <= 65504.f) | !(__builtin_fabsf(u[3]) <= 65504.f);
          } else {
            h01 = __builtin_bit_cast(unsigned, __builtin_convertvector(ce_f32x2{u[0], u[1]}, ce_bf16x2));
            h23 = __builtin_bit_cast(unsigned, __builtin_convertvector(ce_f32x2{u[2], u[3]}, ce_bf16x2));
            const ce_f32x2 r01 = {u[0] - __builtin_bit_cast(float, h01 << 16), u[1] - __builtin_bit_cast(float, h01 & 0xffff0000u)};
            const ce_f32x2 r23 = {u[2] - __builtin_bit_cast(float, h23 << 16), u[3] - __builtin_bit_cast(float, h23 & 0xffff0000u)};
            l01 = __builtin_bit_cast(unsigned, __builtin_convertvector(r01, ce_bf16x2));
            l23 = __builtin_bit_cast(unsigned, __builtin_convertvector(r23, ce_bf16x2));
          }
          // quad_perm [1,0,3,2]: swap with the neighbouring lane (the other half of the 8-channel group)
          const unsigned s0 = even ? l01 : h01, s1 = even ? l23 : h23;
          const unsigned g0 = (unsigned)__builtin_amdgcn_mov_dpp((int)s0, 0xB1, 0xf, 0xf, false);
          const unsigned g1 = (unsigned)__builtin_amdgcn_mov_dpp((int)s1, 0xB1, 0xf, 0xf, false);
          const ce_u32x4 w = even ? ce_u32x4{h01, h23, g0, g1} : ce_u32x4{g0, g1, l01, l23};
          o = __builtin_bit_cast(ce_f32x4, w);
        } else {
          o = u;
        }
        // 16-bit mode: the lo half (odd lanes) is never read by a consumer -- not written either
        if (opix[q] >= 0 && !(SPLIT && f16 && !even)) *(VFX_CE_GLOBAL ce_f32x4*)(p.out_act + (int64_t)opix[q] * astride + aoff) = o;
      }
      if constexpr (SPLIT) {
        if (f16 && __any(f16_sat) && p.flags && (tid & 63) == 0) or_flag_global(p.flags, VFX_FLAG_F16_SATURATED);
      }
    }
  }
}

}  // namespace vfx
