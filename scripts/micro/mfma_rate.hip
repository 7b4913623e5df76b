// mfma_rate.hip -- microbenchmark (round 6): how fast does one SIMD of gfx950 retire v_mfma_f32_32x32x16_bf16 when the stream has
// NACC accumulators in rotation (dependent MFMAs NACC apart) and W waves share the SIMD?  Registers only, no memory traffic.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_rate.hip -o voicefixer_main_amd/abl/mfma_rate && .../mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int NVALU>
__global__ __launch_bounds__(256) void k_rate(int iters, unsigned long long* out, float* sink) {
  extern __shared__ float smem[];
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (i + 1)); }
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
  float v = (float)threadIdx.x;
  const unsigned long long t0 = __builtin_readcyclecounter();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 12 / NACC; ++rep)
#pragma unroll
      for (int n = 0; n < NACC; ++n) {
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NVALU; ++k) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v));
      }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = v;
  for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
  if (s == 12345.678f) sink[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0;
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0;
  }
}

template <int NACC, int NVALU>
static void run(int waves_per_simd, int iters) {
  const int blocks = 256 * waves_per_simd;
  unsigned long long* out;
  float* sink;
  hipMalloc(&out, sizeof(unsigned long long) * blocks * 8);
  hipMalloc(&sink, 1024 * 4);
  const int lds = 160 * 1024 / waves_per_simd - (waves_per_simd > 1 ? 1024 : 0);
  hipFuncSetAttribute((const void*)k_rate<NACC, NVALU>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_rate<NACC, NVALU>), dim3(blocks), dim3(256), lds, 0, iters, out, sink);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_rate<NACC, NVALU>), dim3(blocks), dim3(256), lds, 0, iters, out, sink);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 8);
  hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0;
  for (int i = 0; i < blocks * 4; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
  cyc /= blocks * 4; rt /= blocks * 4;
  const double n = 12.0 * iters;
  printf("NACC=%d VALU/MFMA=%d waves/SIMD=%d: %.1f s_memtime ticks per MFMA and wave, %.1f per MFMA and SIMD; wave %.1f us (100 MHz clock) -> %.0f ticks/us; "
         "kernel %.3f ms -> %.1f ns per MFMA and SIMD\n",
         NACC, NVALU, waves_per_simd, cyc / n, cyc / n / waves_per_simd, rt / 100.0, cyc / (rt / 100.0), ms, ms * 1e6 / (n * waves_per_simd));
  hipFree(out); hipFree(sink);
}

int main() {
  const int iters = 4000;
  for (int w = 1; w <= 3; ++w) {
    run<1, 0>(w, iters);
    run<2, 0>(w, iters);
    run<4, 0>(w, iters);
    run<2, 2>(w, iters);
    run<2, 5>(w, iters);
    run<2, 8>(w, iters);
    run<4, 5>(w, iters);
  }
  return 0;
}
