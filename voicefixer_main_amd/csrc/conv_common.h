// conv_common.h -- device helpers shared by the convolution kernels (conv.hip, resblock.hip).
#pragma once
#include "vfx_internal.h"

namespace vfx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// Four fp16 values (8 bytes: this thread's 4 channels of one row of the fp16 trunk of the 16-bit mode) as ONE vector.
// hipcc 7.2 miscompiles __builtin_bit_cast(T, v[i]) of a vector ELEMENT lvalue: it reads element 0 whatever i is (found on the
// gfx950 assembly in round 4 -- an 8-byte load narrowed to 4 bytes, both halves of a row piece equal); so the halves are never
// picked out of the u32x2 one by one: the whole vector is cast.
__device__ __forceinline__ f32x4 f16x4_widen(u32x2 r) {
  const f16x4 h = __builtin_bit_cast(f16x4, r);
  return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}
// LeakyReLU on the packed halves, max(x, slope x) for 0 < slope <= 1 (v_pk_mul_f16 + v_pk_max_f16): the MFMA operand form of a
// raw fp16 row piece -- no conversion, nothing to saturate
__device__ __forceinline__ u32x2 f16x4_lrelu(u32x2 r, float slope) {
  const f16x4 h = __builtin_bit_cast(f16x4, r);
  const _Float16 s = (_Float16)slope;
  const f16x4 sv = {s, s, s, s};
  return __builtin_bit_cast(u32x2, __builtin_elementwise_max(h, h * sv));
}

// fp16 operand form of the 16-bit mode (TapConvParams::hionly): round to nearest, saturate instead of overflowing
__device__ __forceinline__ unsigned pack_f16x2(float a, float b) {
  // v_med3_f32 = the clamp in one instruction (a NaN comes out as min3 = -65504, exactly what fmin(fmax(x, -65504), 65504) gives)
  const f32x2 v = {__builtin_amdgcn_fmed3f(a, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f)};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}

// One predicate for "this value does not survive the conversion" everywhere (kernels, conv_epilogue.h): not inside
// [-65504, 65504] -- written so that a NaN is flagged too.
__device__ __forceinline__ bool f16_out_of_range(float x) { return !(__builtin_fabsf(x) <= 65504.f); }

// The same, recording whether the clamp changed a value (VFX_FLAG_F16_SATURATED).
__device__ __forceinline__ unsigned pack_f16x2(float a, float b, bool& sat) {
  sat = sat | (bool)((int)f16_out_of_range(a) | (int)f16_out_of_range(b));
  return pack_f16x2(a, b);
}
// The same with the record kept in a VECTOR register: `sat` = max over the converted values of bits(|x|) as an unsigned integer
// (one v_and per value + one v_max3_u32 per pair).  |x| > 65504, infinities and NaNs all have larger bit patterns than 65504.0f
// (0x477fe000), so `f16_sat_bits_bad(sat)` is the predicate of f16_out_of_range over everything converted -- without the
// v_cmp -> s_or chain of the bool form, whose VALU-to-SALU dependency per value made the conversion loops of the 4-wave
// ResStack kernels issue-latency-bound (round 3, phase stamps: 7 k cycles for 80 values per lane).
__device__ __forceinline__ unsigned pack_f16x2(float a, float b, unsigned& sat) {
  const unsigned ua = __builtin_bit_cast(unsigned, a) & 0x7fffffffu, ub = __builtin_bit_cast(unsigned, b) & 0x7fffffffu;
  sat = max(sat, max(ua, ub));
  return pack_f16x2(a, b);
}
__device__ __forceinline__ bool f16_sat_bits_bad(unsigned sat) { return sat > 0x477fe000u; }

// h = LeakyReLU(t) as fp16 MFMA operands, cheaper (round 4: the h-write phase of the fused ResStack kernels was ~40 % VALU issue,
// 14 instructions per pair of values): convert FIRST (saturating, 3 instructions per pair), then activate the packed halves
// (v_pk_mul_f16 + v_pk_max_f16 per PAIR instead of v_mul + v_max per value), and keep the saturation record on the packed result:
// `sat16` = per-half maximum of |h| as two u16 (v_and + v_pk_max_u16 per pair); a half that reached 0x7bff = 65504 was clamped
// (or was exactly 65504 / a NaN: med3 sends a NaN to -65504) -- f16_sat16_bad().  8 instructions per pair with the mask select.
// fp16(LeakyReLU(t)) and LeakyReLU16(fp16(t)) differ by one more rounding of the (100 x smaller) negative values only.
// The position mask goes BETWEEN the conversion and the record: an h position outside the tile / the sequence may have been
// computed from LDS rows that hold anything (the second layer of a pair reads operand rows that the staged fp32 tile overlaid),
// its value is discarded and must not raise the flag (found on the GPU in round 4: the record briefly sat before the mask).
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_f16x2_sat16(float a, float b, bool valid, unsigned& sat16) {
  const unsigned p = valid ? pack_f16x2(a, b) : 0u;
  const u16x2 m = __builtin_bit_cast(u16x2, p & 0x7fff7fffu);
  sat16 = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, sat16), m));
  return p;
}
__device__ __forceinline__ bool f16_sat16_bad(unsigned s) { return (s & 0xffffu) >= 0x7bffu || (s >> 16) >= 0x7bffu; }
__device__ __forceinline__ unsigned lrelu_f16x2(unsigned p, f16x2 slope2) {
  const f16x2 h = __builtin_bit_cast(f16x2, p);
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(h, h * slope2));
}

// One atomic per wave that saw a clamp (rare path).
// The atomic goes through a GLOBAL-address-space pointer on purpose.  atomicOr on a generic `int*` is a flat_atomic_or, and
// the compiler's wait-count insertion treats a FLAT operation as one that may complete out of order on vmcnt AND lgkmcnt:
// until it sees a vmcnt(0) of its own -- never, in kernels whose vmcnt waits are hand-counted inline asm -- it turns every
// later "s_waitcnt lgkmcnt(N)" into lgkmcnt(0), i.e. a software-pipelined fragment read is waited for right after it is
// issued (found on the gfx950 assembly of resblock_r128 / resblock_w64: 59 of 74 LDS waits behind the first report were 0).
__device__ __forceinline__ void or_flag_global(int* flags, int bits) {
  __hip_atomic_fetch_or((__attribute__((address_space(1))) int*)flags, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void report_f16_saturation(bool sat, int* flags) {
  if (__any(sat) && flags && (threadIdx.x & 63) == 0) or_flag_global(flags, VFX_FLAG_F16_SATURATED);
}

// Phase stamps of the timing builds (-DVFX_TIMING; the shipped build compiles them away): lane 0 of every wave keeps
// s_memtime at the phase boundaries and writes them out at the end of the tile.
#ifdef VFX_TIMING
#define VFX_TS_DECL unsigned long long vfx_ts_[16] = {}
#define VFX_TS(i) vfx_ts_[i] = __builtin_readcyclecounter()
#define VFX_TS_FLUSH(ptr, tile, wave, nwaves)                                                          \
  do {                                                                                                 \
    if ((ptr) && (threadIdx.x & 63) == 0)                                                              \
      for (int i_ = 0; i_ < 16; ++i_) (ptr)[((size_t)(tile) * (nwaves) + (wave)) * 16 + i_] = vfx_ts_[i_]; \
  } while (0)
#else
#define VFX_TS_DECL
#define VFX_TS(i)
#define VFX_TS_FLUSH(ptr, tile, wave, nwaves)
#endif

// n / d for the host's r = ceil(2^32 / d) (exact while n * d < 2^32, plan_resblock): r needs 33 bits when d = 1
__device__ __forceinline__ int div_recip(int n, unsigned long long r) {
  return (int)(((unsigned long long)(unsigned)n * (unsigned)r) >> 32) + ((r >> 32) ? n : 0);
}

constexpr int CBM = 128;                     // pixels per tile
constexpr int kOtabSlots = 4;                // k_conv: output pixel tables per block (a block of a phased launch covers up to BN / 32 phases)
constexpr int CROW = 128;                    // bytes per patch row (32 channels)
constexpr int CNQ = kPatchMaxRows / 32;      // patch row groups (one DMA instruction / register group each)
constexpr int CPATCH = kPatchMaxRows * CROW; // bytes per patch buffer

#define VFX_GLOBAL __attribute__((address_space(1)))
#define VFX_LDS __attribute__((address_space(3)))
#define VFX_CONST __attribute__((address_space(4)))

// One weight fragment group = the four 16-byte-per-lane fragments of a (tap, 32-channel chunk, 32 couts).
struct BFrag {
  f32x4 f[4];  // split: (hi, lo) of k 0..15, (hi, lo) of k 16..31; fp32: the four k8 groups
};

template <int N>
struct BGroup {  // the weight fragments of one tap for a wave's N cout blocks (k_conv: N = 1)
  BFrag f[N];
};

// Hidden from the compiler's s_waitcnt bookkeeping on purpose (see the file header): the destination
// registers are only valid after wait_b<N>() with N = number of VMEM operations issued after this load.
__device__ __forceinline__ void load_b_asm(BFrag& R, const float* wtap, unsigned voff) {
  asm volatile(
      "s_nop 4\n\t"
      "global_load_dwordx4 %0, %4, %5\n\t"
      "global_load_dwordx4 %1, %4, %5 offset:1024\n\t"
      "global_load_dwordx4 %2, %4, %5 offset:2048\n\t"
      "global_load_dwordx4 %3, %4, %5 offset:3072"
      : "=&v"(R.f[0]), "=&v"(R.f[1]), "=&v"(R.f[2]), "=&v"(R.f[3])
      : "v"(voff), "s"(wtap)
      : "memory");
}
// 16-bit (fp16) mode: only the hi fragments (f[0], f[2]) of the group.
__device__ __forceinline__ void load_b_asm_hi(BFrag& R, const float* wtap, unsigned voff) {
  asm volatile(
      "s_nop 4\n\t"
      "global_load_dwordx4 %0, %2, %3\n\t"
      "global_load_dwordx4 %1, %2, %3 offset:2048"
      : "=&v"(R.f[0]), "=&v"(R.f[2])
      : "v"(voff), "s"(wtap)
      : "memory");
}
template <int N>
__device__ __forceinline__ void wait_b(BFrag& R) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(R.f[0]), "+v"(R.f[1]), "+v"(R.f[2]), "+v"(R.f[3]) : "n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);  // nothing that reads R may be scheduled above the wait
}

// The group's registers are readable from here on (the s_waitcnt that made them so precedes this in program order:
// asm volatile statements are not reordered); nothing that reads R may be scheduled above it.
__device__ __forceinline__ void use_b(BFrag& R) {
  asm volatile("" : "+v"(R.f[0]), "+v"(R.f[1]), "+v"(R.f[2]), "+v"(R.f[3]) : : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void use_b_hi(BFrag& R) {
  asm volatile("" : "+v"(R.f[0]), "+v"(R.f[2]) : : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// Logical patch-row layout: a row is 128 bytes = 8 pieces of 16 bytes (split-bf16: pieces 0..3 = 32 hi bf16,
// 4..7 = 32 lo bf16; fp32: 32 floats).  Piece p of LDS row r sits at slot p ^ ((r >> 1) & 7).
__device__ __forceinline__ int swz_key(int row) { return ((row >> 1) & 7) << 4; }

}  // namespace vfx
