// store_pattern.hip -- microbenchmark (round 6): write bandwidth of 16-byte buffer stores by lane pattern.
//   A: full lines   -- 8 consecutive lanes write one 128-byte row (a wave-instruction covers 8 whole rows);
//   B: MFMA layout  -- lane (l31, lh) writes bytes [32 j + 16 lh, +16) of row l31: an instruction touches 32 rows with 32 bytes each,
//                      four instructions (j = 0 .. 3) complete the 32 rows (what the direct epilogues of the fused kernels do);
//   C: the same with the four instructions of a row group issued back to back per group (B issues j-major over two row groups).
// Every wave writes `rows_per_wave` rows of 128 bytes; 256 threads per block.
//   hipcc --offload-arch=gfx950 -O3 -w scripts/micro/store_pattern.hip -o voicefixer_main_amd/abl/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(256) void k_store(char* out, int rows_per_wave, long long total_rows) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long row0 = wave * rows_per_wave;
  if (row0 >= total_rows) return;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out + row0 * 128, 0, rows_per_wave * 128, 0x00020000);
  u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
  if (PAT == 0) {
    const int lr = lane >> 3, cg = lane & 7;
    for (int g = 0; g < rows_per_wave; g += 8) {
      __builtin_amdgcn_raw_buffer_store_b128(v, r, (g + lr) * 128 + cg * 16, 0, 0);
      v[1] += 1u;
    }
  } else {
    const int l31 = lane & 31, lh = lane >> 5;
    for (int g = 0; g < rows_per_wave; g += 32) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        __builtin_amdgcn_raw_buffer_store_b128(v, r, (g + l31) * 128 + 32 * j + 16 * lh, 0, 0);
        v[1] += 1u;
        if (PAT == 2) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");  // spaced out, as between MFMAs
      }
    }
  }
}

template <int PAT>
static void run(const char* name, char* buf, long long bytes) {
  const int rows_per_wave = 512;
  const long long total_rows = bytes / 128;
  const int blocks = (int)((total_rows / rows_per_wave + 3) / 4);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_store<PAT>), dim3(blocks), dim3(256), 0, 0, buf, rows_per_wave, total_rows);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_store<PAT>), dim3(blocks), dim3(256), 0, 0, buf, rows_per_wave, total_rows);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %.3f ms per %.2f GB -> %.2f TB/s\n", name, ms / 5, bytes / 1e9, bytes / 1e9 / (ms / 5));
}

int main() {
  const long long bytes = 1LL << 30;
  char* buf;
  hipMalloc(&buf, bytes);
  run<0>("A full lines (8 lanes per 128-byte row)", buf, bytes);
  run<1>("B MFMA layout (32 rows x 32 bytes per store)", buf, bytes);
  run<2>("C MFMA layout, stores spaced by 128 cycles", buf, bytes);
  run<0>("A again", buf, bytes);
  hipFree(buf);
  return 0;
}
