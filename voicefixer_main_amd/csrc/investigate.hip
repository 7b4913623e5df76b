// investigate.hip -- instruments of the round-5 two-stream investigation (profiles/r05_two_streams.md).  NOT product code: this
// translation unit is linked into libvfx_test.so only (csrc/Makefile), nothing in libvfx.so refers to it.
//   * vfx_debug_burn: synthetic co-runners for the other stream (MFMA / VALU / LDS-DMA / store / sleep loops);
//   * vfx_debug_voc_final: the vocoder tail (small_ops.hip: k_voc_final, C = 32) with every partial sum kept TWICE in separate
//     registers; lanes whose copies differ -- before the group reduction (code 1) or after it (code 2) -- are logged in
//     g_vf_dbg (vfx_debug_read_vf / vfx_debug_reset_vf).  `lds_bytes` > 0 asks for that much dynamic LDS the kernel never touches
//     (no convolution block fits on a CU beside it).
#include "conv_common.h"
#include "vfx_internal.h"

namespace vfx {

// Round-5 investigation build (scripts/two_streams_registers.py): every sum is kept twice, in separate registers; lanes whose copies
// differ -- before the group reduction (code 1) or after it (code 2) -- are logged here.
__device__ unsigned g_vf_dbg[1 + 8 * 4096];
extern "C" int vfx_debug_read_vf(unsigned* out, int n_words) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vf_dbg), sizeof(unsigned) * (size_t)n_words, 0, hipMemcpyDeviceToHost);
}
extern "C" int vfx_debug_reset_vf() {
  unsigned zero = 0;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_vf_dbg), &zero, sizeof(zero), 0, hipMemcpyHostToDevice);
}
// Synthetic co-runners for the other stream (scripts/two_streams_burners.py): what kind of work disturbs k_voc_final?
//   0: 16-bit MFMAs on registers only (v_mfma_f32_32x32x16_bf16, four independent accumulators per wave), no memory traffic
//   1: the same with fp32 MFMAs (v_mfma_f32_32x32x2f32)        2: VALU FMAs only
//   3: 16-byte LDS-DMA reads of a buffer (buffer_load ... lds, the convolutions' patch path), no arithmetic
typedef unsigned ce_u32x4_burn __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(256) void k_burn(int iters, const float* __restrict__ src, unsigned src_bytes, float* __restrict__ out) {
  const int tid = threadIdx.x;
  float keep = 0.f;
  if constexpr (KIND == 0) {
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      a[e] = (__bf16)(1.0f + 0.001f * (float)((tid + e) & 31));
      b[e] = (__bf16)(0.5f - 0.002f * (float)((tid * 3 + e) & 15));
    }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, b, c3, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) keep += c0[r] + c1[r] + c2[r] + c3[r];
  } else if constexpr (KIND == 1) {
    const float a = 1.0f + 0.001f * (float)(tid & 31), b = 0.5f - 0.002f * (float)(tid & 15);
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, c3, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) keep += c0[r] + c1[r] + c2[r] + c3[r];
  } else if constexpr (KIND == 2) {
    float x0 = 1.0f + 0.001f * (float)tid, x1 = 0.5f, x2 = 0.25f, x3 = 0.125f;
    const float m = 0.999f, d = 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        x0 = fmaf(x0, x1, d);   // cross-coupled: no closed form for the compiler to fold the loop into
        x1 = fmaf(x1, x2, m);
        x2 = fmaf(x2, x3, d);
        x3 = fmaf(x3, x0, m);
      }
    }
    keep = x0 + x1 + x2 + x3;
  } else if constexpr (KIND == 6) {
    // 10 000 different VALU instructions in a row (~100 KB of code): every CU running it keeps replacing the instruction cache
    // lines it shares with its neighbours -- does the victim only have to MISS in the instruction cache?
    float x0 = 1.0f + 0.001f * (float)tid, x1 = 0.999f;
#define VFX_B1(i) x0 = fmaf(x0, x1, 1.0f + (float)(i) * 1.0e-6f);
#define VFX_B10(b) VFX_B1(b) VFX_B1(b + 1) VFX_B1(b + 2) VFX_B1(b + 3) VFX_B1(b + 4) VFX_B1(b + 5) VFX_B1(b + 6) VFX_B1(b + 7) VFX_B1(b + 8) VFX_B1(b + 9)
#define VFX_B100(b) VFX_B10(b) VFX_B10(b + 10) VFX_B10(b + 20) VFX_B10(b + 30) VFX_B10(b + 40) VFX_B10(b + 50) VFX_B10(b + 60) VFX_B10(b + 70) VFX_B10(b + 80) VFX_B10(b + 90)
#define VFX_B1000(b) VFX_B100(b) VFX_B100(b + 100) VFX_B100(b + 200) VFX_B100(b + 300) VFX_B100(b + 400) VFX_B100(b + 500) VFX_B100(b + 600) VFX_B100(b + 700) VFX_B100(b + 800) VFX_B100(b + 900)
    for (int i = 0; i < iters; ++i) {
      VFX_B1000(0) VFX_B1000(1000) VFX_B1000(2000) VFX_B1000(3000) VFX_B1000(4000)
      VFX_B1000(5000) VFX_B1000(6000) VFX_B1000(7000) VFX_B1000(8000) VFX_B1000(9000)
      x1 = 0.999f + 1e-9f * x0;
    }
    keep = x0;
  } else if constexpr (KIND == 7 || KIND == 8) {
    // 7: 16-byte streaming stores only.  8: a convolution's loop in miniature -- LDS-DMA reads, a wait, a block barrier, 16-bit MFMAs
    // on fragments read from LDS, 16-byte stores
    extern __shared__ float burn_lds[];
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)src_bytes, 0x00020000);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned off = ((unsigned)blockIdx.x * 4096u + (unsigned)tid * 16u) % (src_bytes - 65536u);
    f32x16 c0 = {}, c1 = {};
    f32x4 val = {1.0f + 0.001f * (float)tid, 2.0f, 3.0f, 4.0f};
    for (int i = 0; i < iters; ++i) {
      if constexpr (KIND == 8) {
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          __attribute__((address_space(3))) void* l = (__attribute__((address_space(3))) void*)((char*)burn_lds + (q * 4 + wave) * 1024);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, (int)(off + (unsigned)q * 8192u), 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 6; ++t) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>((const char*)burn_lds + t * 4096 + (tid & 255) * 16);
          const bf16x8 b = *reinterpret_cast<const bf16x8*>((const char*)burn_lds + t * 4096 + ((tid + 64) & 255) * 16);
          c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
        }
        __syncthreads();
        val[0] = c0[0] + c1[1];
      }
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ce_u32x4_burn, val), rsrc, (int)off, 0, 0);
      off = (off + 1048576u + 49152u) % (src_bytes - 65536u);
      off &= ~15u;
    }
    keep = c0[3] + c1[5] + val[1];
  } else if constexpr (KIND == 4) {
    // occupies wave slots (KIND 4: nothing else; launched with 52 KB of LDS as kind 5: three blocks fill a CU's LDS like the
    // convolutions do) and sleeps: no arithmetic, no memory traffic -- does the victim only have to WAIT for resources?
    for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(64);
    keep = (float)iters;
  } else {
    extern __shared__ float burn_lds[];
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)src_bytes, 0x00020000);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned off = ((unsigned)blockIdx.x * 4096u + (unsigned)tid * 16u) % (src_bytes - 65536u);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        __attribute__((address_space(3))) void* l = (__attribute__((address_space(3))) void*)((char*)burn_lds + (q * 4 + wave) * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, (int)(off + (unsigned)q * 8192u), 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
      off = (off + 1048576u + 49152u) % (src_bytes - 65536u);
      off &= ~15u;
    }
    __syncthreads();
    keep = burn_lds[tid];
  }
  if (keep == 12345.678f) out[tid] = keep;  // keeps the work live
}
extern "C" int vfx_debug_burn(int kind, int blocks, int iters, const float* src, unsigned src_bytes, float* out, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (kind) {
    case 0: hipLaunchKernelGGL(k_burn<0>, dim3(blocks), dim3(256), 0, s, iters, src, src_bytes, out); break;
    case 1: hipLaunchKernelGGL(k_burn<1>, dim3(blocks), dim3(256), 0, s, iters, src, src_bytes, out); break;
    case 2: hipLaunchKernelGGL(k_burn<2>, dim3(blocks), dim3(256), 0, s, iters, src, src_bytes, out); break;
    case 3: hipLaunchKernelGGL(k_burn<3>, dim3(blocks), dim3(256), 24576, s, iters, src, src_bytes, out); break;
    case 4: hipLaunchKernelGGL(k_burn<4>, dim3(blocks), dim3(256), 0, s, iters, src, src_bytes, out); break;
    case 6: hipLaunchKernelGGL(k_burn<6>, dim3(blocks), dim3(256), 0, s, iters, src, src_bytes, out); break;
    case 7: hipLaunchKernelGGL(k_burn<7>, dim3(blocks), dim3(256), 24576, s, iters, src, src_bytes, out); break;
    case 8: hipLaunchKernelGGL(k_burn<8>, dim3(blocks), dim3(256), 24576, s, iters, src, src_bytes, out); break;
    default: {
      static bool once = [] {
        VFX_HIP(hipFuncSetAttribute((const void*)k_burn<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 53248));
        return true;
      }();
      (void)once;
      hipLaunchKernelGGL(k_burn<4>, dim3(blocks), dim3(256), 53248, s, iters, src, src_bytes, out);
      break;
    }
  }
  return (int)hipGetLastError();
}
__device__ __forceinline__ void vf_log(unsigned code, int o, float a, float b) {
  const unsigned slot = atomicAdd(&g_vf_dbg[0], 1u);
  if (slot < 4096u) {
    unsigned* w = &g_vf_dbg[1 + 8 * slot];
    w[0] = code; w[1] = blockIdx.x; w[2] = blockIdx.y; w[3] = threadIdx.x; w[4] = (unsigned)o;
    w[5] = __float_as_uint(a); w[6] = __float_as_uint(b); w[7] = 0;
  }
}

// k_voc_final<4, X16> with doubled sums (see small_ops.hip for the product kernel and its comments)
template <bool X16>
__global__ __launch_bounds__(256) void k_voc_final_dbg(const float* __restrict__ x, int T, const float* __restrict__ w, float bias,
                                                        float slope, float* __restrict__ wav, const int* __restrict__ lens, int hop) {
  constexpr int CPL = 4, C = 32;
  const int b = blockIdx.y;
  const int Ts = T;
  if (lens) T = min(T, lens[b] * hop);
  const int grp = threadIdx.x >> 3, g = threadIdx.x & 7;
  const int t0 = (blockIdx.x * 32 + grp) * 8;
  const float* xb = x + (int64_t)b * Ts * C + g * CPL;
  const _Float16* xh = reinterpret_cast<const _Float16*>(x) + (int64_t)b * Ts * C + g * CPL;
  float wk[7][CPL];
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(w + k * C + g * CPL);
#pragma unroll
    for (int e = 0; e < 4; ++e) wk[k][e] = v[e];
  }
  float acc[8], acc2[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) acc[o] = acc2[o] = 0.f;
  if (t0 < T) {
#pragma unroll
    for (int r = 0; r < 14; ++r) {
      int tt = t0 - 3 + r;
      tt = tt < 0 ? -tt : tt;
      tt = tt >= T ? 2 * (T - 1) - tt : tt;
      tt = tt < 0 ? 0 : tt;
      f32x4 a;
      if constexpr (X16) {
        const f16x4 hv = *reinterpret_cast<const f16x4*>(xh + (int64_t)tt * C);
        a = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
      } else {
        a = *reinterpret_cast<const f32x4*>(xb + (int64_t)tt * C);
      }
      float v[CPL];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = a[e] >= 0.f ? a[e] : a[e] * slope;
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        const int k = r - o;
        if (k >= 0 && k < 7) {
#pragma unroll
          for (int c = 0; c < CPL; ++c) acc[o] = fmaf(v[c], wk[k][c], acc[o]);
#pragma unroll
          for (int c = 0; c < CPL; ++c) {
            float vv = v[c];
            asm volatile("" : "+v"(vv));
            acc2[o] = fmaf(vv, wk[k][c], acc2[o]);
          }
        }
      }
    }
  }
  float mine = 0.f, mine2 = 0.f;
#pragma unroll
  for (int o = 0; o < 8; ++o)
    if (__float_as_uint(acc[o]) != __float_as_uint(acc2[o])) vf_log(1u, o, acc[o], acc2[o]);
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    float s = acc[o];
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    mine = g == o ? s : mine;
    float s2 = acc2[o];
    asm volatile("" : "+v"(s2));
    s2 += __shfl_xor(s2, 1);
    s2 += __shfl_xor(s2, 2);
    s2 += __shfl_xor(s2, 4);
    mine2 = g == o ? s2 : mine2;
  }
  if (__float_as_uint(mine) != __float_as_uint(mine2)) vf_log(2u, g, mine, mine2);
  const int t = t0 + g;
  if (t < T) wav[(int64_t)b * Ts + t] = tanhf(mine + bias);
}

extern "C" int vfx_debug_voc_final(const float* x, int x_f16, int B, int T, const float* w, float bias, float slope, float* wav,
                                   const int* lens, int hop, int lds_bytes, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((T + 255) / 256, B);
  if (lds_bytes > 0) {
    (void)hipFuncSetAttribute((const void*)k_voc_final_dbg<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    (void)hipFuncSetAttribute((const void*)k_voc_final_dbg<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  }
  if (x_f16) hipLaunchKernelGGL(k_voc_final_dbg<true>, grid, dim3(256), (size_t)lds_bytes, s, x, T, w, bias, slope, wav, lens, hop);
  else hipLaunchKernelGGL(k_voc_final_dbg<false>, grid, dim3(256), (size_t)lds_bytes, s, x, T, w, bias, slope, wav, lens, hop);
  return (int)hipGetLastError();
}

// Where do the blocks of a launch on `stream` run?  out[2048]: histogram over (XCC_ID << 8 | SE_ID << 5 | SH_ID << 4 | CU_ID) of the
// blocks' first waves -- validates a CU mask (hipExtStreamCreateWithCUMask) before an experiment relies on it.
__global__ __launch_bounds__(64) void k_where(unsigned* __restrict__ out, int spin) {
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(32);   // keep the block resident so that the launch spreads over the CUs it may use
  if (threadIdx.x == 0) atomicAdd(out + (((xcc & 15u) << 8) | (((hw >> 13) & 7u) << 5) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u)), 1u);
}
extern "C" int vfx_debug_where(unsigned* out2048, int blocks, int spin, void* stream) {
  hipLaunchKernelGGL(k_where, dim3(blocks), dim3(64), 0, static_cast<hipStream_t>(stream), out2048, spin);
  return (int)hipGetLastError();
}

}  // namespace vfx
