// conv_epilogue.h -- shared epilogue of k_tapconv / k_patchconv.
//
// The 32x32 MFMA accumulator fragments go through LDS (re-using the staging buffers) so that the
// residual read and the output write are 16-byte-per-lane, row-contiguous accesses.  All residual
// loads of a thread are issued first, all results are formed in registers, and only then are the
// stores issued back-to-back: on gfx950 vmcnt also counts stores, so a store placed between two
// waited loads would serialise on the previous store's acknowledgement.
#pragma once
#include "vfx_internal.h"

namespace vfx {

typedef float ce_f32x4 __attribute__((ext_vector_type(4)));
typedef float ce_f32x16 __attribute__((ext_vector_type(16)));
#define VFX_CE_GLOBAL __attribute__((address_space(1)))

template <int BN, int WM, int WN, int WAVES_N>
__device__ __forceinline__ void conv_epilogue(const TapConvParams& p, float* smem, const int* otab,
                                              ce_f32x16 (&acc)[WM][WN], int n0) {
  constexpr int LDO = BN + 4;  // staged row length (floats), keeps 16-byte alignment
  constexpr int V = BN / 4;    // float4 per output row
  constexpr int RPP = 256 / V; // rows per pass
  constexpr int NPASS = 128 / RPP;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int l31 = lane & 31, lh = lane >> 5;
  __syncthreads();  // every wave is done reading the last K step
  // C/D layout of the 32x32 MFMA: col = lane & 31 (-> cout), row = (r&3) + 8*(r>>2) + 4*(lane>>5).
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int b = 0; b < WN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wm * WM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        smem[row * LDO + (wn * WN + b) * 32 + l31] = acc[a][b][r];
      }
  __syncthreads();
  const int c4 = tid % V, r0 = tid / V;
  const int Cout = p.Cout;
  const int ncol = n0 + 4 * c4;
  ce_f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) bv = *(const VFX_CE_GLOBAL ce_f32x4*)(p.bias + ncol);
  int opix[NPASS];
  ce_f32x4 val[NPASS];
#pragma unroll
  for (int q = 0; q < NPASS; ++q) {
    opix[q] = otab[r0 + q * RPP];
    val[q] = *reinterpret_cast<const ce_f32x4*>(smem + (r0 + q * RPP) * LDO + 4 * c4) + bv;
  }
  if (p.residual) {
    ce_f32x4 res[NPASS];
#pragma unroll
    for (int q = 0; q < NPASS; ++q)
      res[q] = *(const VFX_CE_GLOBAL ce_f32x4*)(p.residual + (int64_t)(opix[q] < 0 ? 0 : opix[q]) * Cout + ncol);
#pragma unroll
    for (int q = 0; q < NPASS; ++q) val[q] += res[q];
  }
#pragma unroll
  for (int q = 0; q < NPASS; ++q)
    if (opix[q] >= 0) *(VFX_CE_GLOBAL ce_f32x4*)(p.out + (int64_t)opix[q] * Cout + ncol) = val[q];
}

}  // namespace vfx
