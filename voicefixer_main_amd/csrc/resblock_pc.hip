// resblock_pc.hip -- the fused wide ResStack layer of the 16-bit mode (resblock_act.hip: C = 256, two-form trunk) as a
// PERSISTENT producer / consumer kernel.
//
//     y  = x + conv2(LeakyReLU(conv1(xa) + b1)) + b2,      ya = fp16(LeakyReLU_next(y))
//
// Why: the one-tile-per-block kernel is the SUM of its memory phases and its arithmetic (timing-only ablation,
// profiles/r02_ablation_resblock_act.txt: 0.50 ms + 0.48 ms = 0.975 ms per layer).  One block per CU cannot overlap the two,
// and a wave cannot either -- vmcnt retires its loads AND stores in order, so a wave that issued the epilogue's stores or
// the next patch request would wait for them at its next weight fetch.  Here the three kinds of memory traffic belong to
// three kinds of waves, each with its own counters, and a block walks a contiguous range of tiles:
//   waves 0..7   MFMA waves: conv1, h, conv2, staging of the accumulators; their only VMEM operations are the weight
//                fragments (L2-resident, hand-counted vmcnt as in k_conv);
//   wave  8      loader: requests the NEXT tile's whole xa patch (4 chunks x 24 KB, LDS-DMA) as soon as conv1 of the
//                current tile has finished reading the patch region, and waits for it while the MFMA waves run conv2 and
//                the epilogue passes;
//   waves 9..11  store waves: fetch the raw residual of a 64-channel pass one pass ahead, add it to the staged pass, write
//                y (fp32) and ya (fp16) -- while the MFMA waves stage the next pass / start the next tile's conv1.
// Synchronisation is s_barrier only (never a spin): every wave of the block passes the same seven barriers per tile,
// whatever its role and whether or not its tile exists, so the kernel cannot hang on a lost signal.
//
// LDS (all 160 KB of the CU): [0, 96 KB) the patch region, [96, 160 KB) h (128 x 512 B) during conv2, then two staging
// buffers A / B of one 64-channel pass each (128 rows x 256 B, 16-byte pieces XOR-swizzled by the row).
// Phases of tile i (| = barrier):
//   P0 conv1(i)              ST: pass 3 of tile i-1 from B          |
//   P1 write h(i)            LD: request patch(i+1)                 |
//   P2 conv2(i)              ST: indices of tile i, residual pass 0 |
//   P3 stage pass 0 -> A                                            |
//   P4 stage pass 1 -> B     ST: pass 0 from A (+ residual pass 1)  |
//   P5 stage pass 2 -> A     ST: pass 1 from B (+ residual pass 2)  |
//   P6 stage pass 3 -> B     ST: pass 2 from A (+ residual pass 3)   LD: patch(i+1) has landed |
#include "conv_common.h"
#include "vfx_internal.h"

namespace vfx {

// lgkmcnt(0): this wave's LDS writes are done before anybody passes the barrier.  No vmcnt: the loader's DMA, the MFMA
// waves' weight fetches and the store waves' stores stay in flight across it.
#define VFX_PC_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <int C>
__global__ __launch_bounds__(768, 3) void k_resblock_pc(const ResBlockParams* __restrict__ pp, int ntiles, int per_block) {
  constexpr int MT = 128;                    // h positions per tile
  constexpr int NCH = C / 64;                // 64-channel chunks (128-byte rows of fp16)
  constexpr int NMW = C / 32;                // MFMA waves: one per 32 output channels
  static_assert(NMW == 8, "8 MFMA waves + loader + 3 store waves = 12 waves = 3 per SIMD");
  constexpr int WM = MT / 32;                // 32-row MFMA blocks per wave
  constexpr int WL = 4;                      // weight loads per tap and wave
  constexpr int HROW = C * 2;                // bytes per h row
  constexpr int NT1 = 3 * NCH;               // taps of conv1 (chunk-major); conv2 has as many
  constexpr int NEP = C / 64;                // epilogue passes of 64 channels
  constexpr int PATCH_BYTES = NCH * CPATCH;  // 96 KB
  constexpr int STG_BYTES = MT * 256;        // one staged pass: 128 rows x 64 floats
  constexpr int NSL = 3 * 64;                // store lanes
  constexpr int NF = (MT * 16 + NSL - 1) / NSL;  // float4 per store lane and pass (11; the last one partial)

  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* const lds = reinterpret_cast<char*>(smem);
  char* const hs = lds + PATCH_BYTES;        // h / staging region

  const ResBlockParams& p = *pp;
  const int tid = threadIdx.x;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const bool is_ld = wave_u == NMW, is_st = wave_u > NMW;  // else: an MFMA wave
  const int T = p.T, d = p.dil, W1 = p.W1, TH = p.TH, PW = p.PW, P = p.P;
  const int rowstride = p.fold ? d : 0;
  const int tiles_per_img = p.tiles_w * p.tiles_h;
  const float slope = p.slope;

  // tile t of this block: a contiguous range, so that a tile's halo rows are in this XCD's L2 from its predecessor
  const int t_begin = blockIdx.x * per_block;
  auto tile_geom = [&](int t, int& img, int& j0, int& base_h) __attribute__((always_inline)) {
    const int tj = t % p.tiles_w;
    const int ti = (t / p.tiles_w) % p.tiles_h;
    img = t / tiles_per_img;
    j0 = tj * p.TWo;
    base_h = p.fold ? ti * TH * d + j0 - 1 : j0 - 1;
  };

  // ---- loader ------------------------------------------------------------------------------------------------------
  // one wave covers 8 patch rows per instruction (lane = row & 7 | slot): 24 instructions per chunk
  auto request_patch = [&](int t) __attribute__((always_inline)) {
    int img, j0, base_h;
    tile_geom(t, img, j0, base_h);
    const int base_x = base_h - d;
    const int cg = lane & 7;
    for (int c = 0; c < NCH; ++c) {
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(p.xa + c * kKC), 0, (int)(unsigned)((int64_t)p.B * T * C * 2 - (int64_t)c * kKC * 4), 0x00020000);
      int prow = lane >> 3;
      int pi = prow / PW, pj = prow - pi * PW;
#pragma unroll 4
      for (int i = 0; i < kPatchMaxRows / 8; ++i) {
        const int pos = base_x + pi * rowstride + pj;
        const bool ok = (prow < P) & ((unsigned)pos < (unsigned)T);
        const int key = (prow >> 1) & 7;
        const unsigned o = ok ? (unsigned)(img * T + pos) * (unsigned)(C * 2) + ((unsigned)(cg ^ key) << 4) : 0xfffffff0u;
        VFX_LDS void* l = (VFX_LDS void*)(lds + c * CPATCH + (8 * i) * CROW);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, (int)o, 0, 0, 0);
        prow += 8;
        pj += 8;
        if (pj >= PW) {  // PW >= 18 > 8: at most one wrap per step
          pj -= PW;
          pi += 1;
        }
      }
    }
  };

  // ---- MFMA waves: per-lane state -----------------------------------------------------------------------------------
  const int wn = wave_u & (NMW - 1);
  int l31 = lane & 31, lh = lane >> 5;  // re-made opaque per tile (below)
  const unsigned nb_off = (unsigned)(wn * 1024 + lane * 4) * 4u;
  const int64_t ts = (int64_t)C * kKC;
  f32x16 acc[WM];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  int arow1[WM];
  bool hval[WM];
  BFrag R0 = {}, R1 = {};
  auto ring = [&](int g) __attribute__((always_inline)) -> BFrag& { return (g & 1) ? R1 : R0; };
  auto fetch = [&](int g) __attribute__((always_inline)) {
    const float* w = g < NT1 ? p.w1 + g * ts : (g < 2 * NT1 ? p.w2 + (g - NT1) * ts : p.w2 + (NT1 - 1) * ts);
    load_b_asm(ring(g), w, nb_off);
  };
  auto wait_tap = [&](int g) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(WL) : "memory");
    use_b(ring(g));
  };
  auto drain = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    use_b(R0);
    use_b(R1);
  };
  auto mma = [&](const BFrag& R, const char* img_base, int stride, const int (&row)[WM], int chunk) __attribute__((always_inline)) {
    const char* base[WM];
    int key[WM];
#pragma unroll
    for (int a = 0; a < WM; ++a) {
      base[a] = img_base + row[a] * stride;
      if (chunk >= 0) base[a] += (chunk ^ (row[a] & 1)) * CROW;  // h: chunk parity swap (resblock.hip)
      key[a] = swz_key(row[a]);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const f16x8 bh = __builtin_bit_cast(f16x8, R.f[s]);
      f16x8 ah[WM];
#pragma unroll
      for (int a = 0; a < WM; ++a) ah[a] = *reinterpret_cast<const f16x8*>(base[a] + ((32 * s + 16 * lh) ^ key[a]));
#pragma unroll
      for (int a = 0; a < WM; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah[a], acc[a], 0, 0, 0);
    }
  };
  // accumulators of this wave's 32 channels -> staging buffer `buf` (the wave's pass only)
  auto stage = [&](char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < WM; ++a) {
      const int row = a * 32 + l31;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int piece = (wn & 1) * 8 + 2 * j + lh;  // 16-byte piece of the 64-channel row: channels (wn&1)*32 + 8j + 4lh ..
        *reinterpret_cast<f32x4*>(buf + row * 256 + ((piece ^ (row & 15)) << 4)) =
            f32x4{acc[a][4 * j], acc[a][4 * j + 1], acc[a][4 * j + 2], acc[a][4 * j + 3]};
      }
    }
  };

  // ---- store waves: per-lane state ----------------------------------------------------------------------------------
  int sl = (wave_u - NMW - 1) * 64 + lane;  // store lane 0..191 (meaningful for store waves only)
  int opix[NF];      // output pixel of this lane's float4 #i (row (sl + 192 i) >> 4), or -1
  f32x4 res[NF];     // the raw residual of the pass that is drained next
  auto st_indices = [&](int t) __attribute__((always_inline)) {
    int img, j0, base_h;
    tile_geom(t, img, j0, base_h);
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const int f = sl + NSL * i;
      const int m = f >> 4;  // staged row = h pixel
      const int li = m / W1, lj = m - li * W1;
      const int pos = base_h + li * rowstride + lj;
      const bool ok = (f < MT * 16) & (li < TH) & (lj >= 1) & (lj <= W1 - 2) & ((unsigned)pos < (unsigned)T) &
                      (!p.fold | (j0 + lj - 1 < d));
      opix[i] = ok ? img * T + pos : -1;
    }
  };
  auto st_prefetch = [&](int pass) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const int c4 = (sl + NSL * i) & 15;
      const int op = opix[i];
      res[i] = *(const VFX_GLOBAL f32x4*)(p.x + (int64_t)(op < 0 ? 0 : op) * C + pass * 64 + 4 * c4);
    }
  };
  auto st_drain = [&](const char* buf, int pass) __attribute__((always_inline)) {
    const bool even = (lane & 1) == 0;
    const float aslope = p.act_slope;
    bool f16_sat = false;
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const int f = sl + NSL * i;
      const int row = (f >> 4) & (MT - 1), c4 = f & 15;
      const int ncol = pass * 64 + 4 * c4;
      const f32x4 bv = *(const VFX_GLOBAL f32x4*)(p.b2 + ncol);
      const int op = opix[i];
      const f32x4 val = *reinterpret_cast<const f32x4*>(buf + row * 256 + ((c4 ^ (row & 15)) << 4)) + bv + res[i];
      if (op >= 0) *(VFX_GLOBAL f32x4*)(p.y + (int64_t)op * C + ncol) = val;
      if (p.ya) {
        f32x4 u;
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = fmaxf(val[e], val[e] * aslope);
        const unsigned h01 = pack_f16x2(u[0], u[1], f16_sat), h23 = pack_f16x2(u[2], u[3], f16_sat);
        // quad_perm [1,0,3,2]: the even lane of a pair collects the pair's 8 consecutive channels (16 bytes)
        const unsigned g0 = (unsigned)__builtin_amdgcn_mov_dpp((int)h01, 0xB1, 0xf, 0xf, false);
        const unsigned g1 = (unsigned)__builtin_amdgcn_mov_dpp((int)h23, 0xB1, 0xf, 0xf, false);
        const u32x4 w = {h01, h23, g0, g1};
        if (op >= 0 && even)
          *(VFX_GLOBAL f32x4*)(p.ya + (int64_t)op * (C / 2) + (ncol >> 1)) = __builtin_bit_cast(f32x4, w);
      }
    }
    if (p.ya) report_f16_saturation(f16_sat, p.flags);
  };

  // ---- the three roles: separate loops (separate register allocation), the SAME barriers: 1 + 7 per tile --------------
  const int n_mine = min(per_block, ntiles - t_begin);  // >= 1 (launch_resblock_pc); block-uniform
  if (is_ld) {
    request_patch(t_begin);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VFX_PC_BAR();
    for (int it = 0; it < n_mine; ++it) {
      VFX_PC_BAR();                                    // P0
      if (it + 1 < n_mine) request_patch(t_begin + it + 1);
      VFX_PC_BAR();                                    // P1
      VFX_PC_BAR();                                    // P2
      VFX_PC_BAR();                                    // P3
      VFX_PC_BAR();                                    // P4
      VFX_PC_BAR();                                    // P5
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next tile's patch has landed
      VFX_PC_BAR();                                    // P6
    }
  } else if (is_st) {
    VFX_PC_BAR();
    for (int it = 0; it < n_mine; ++it) {
      asm volatile("" : "+v"(sl));                     // keeps the per-lane addresses inside the loop
      if (it > 0) st_drain(hs + STG_BYTES, NEP - 1);
      VFX_PC_BAR();                                    // P0
      VFX_PC_BAR();                                    // P1
      st_indices(t_begin + it);
      st_prefetch(0);
      VFX_PC_BAR();                                    // P2
      VFX_PC_BAR();                                    // P3
#pragma unroll
      for (int pass = 1; pass < NEP; ++pass) {
        st_drain(hs + ((pass - 1) & 1) * STG_BYTES, pass - 1);
        st_prefetch(pass);                             // fetched while the MFMA waves stage that pass
        VFX_PC_BAR();                                  // P4 .. P6
      }
    }
    st_drain(hs + STG_BYTES, NEP - 1);                 // nobody overwrites B any more
  } else {
    fetch(0);
    VFX_PC_BAR();
    for (int it = 0; it < n_mine; ++it) {
      asm volatile("" : "+v"(l31), "+v"(lh));          // the tile-independent LDS addresses stay inside the loop (registers)
      // ---- P0: conv1 ----
      {
        int img, j0, base_h;
        tile_geom(t_begin + it, img, j0, base_h);
#pragma unroll
        for (int a = 0; a < WM; ++a) {
          const int ml = a * 32 + l31;
          const int li = ml / W1, lj = ml - li * W1;
          arow1[a] = li < TH ? li * PW + lj : 0;
          const int pos = base_h + li * rowstride + lj;
          hval[a] = (li < TH) & ((unsigned)pos < (unsigned)T);
        }
      }
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int g = 3 * c + k;
          fetch(g + 1);
          wait_tap(g);
          int rows[WM];
#pragma unroll
          for (int a = 0; a < WM; ++a) {
            int r = arow1[a];
            asm volatile("" : "+v"(r));  // per-tap addresses are recomputed, not kept: 12 taps x 4 rows x 5 registers otherwise
            rows[a] = r + p.poff[k];
          }
          mma(ring(g), lds + c * CPATCH, CROW, rows, -1);
          __builtin_amdgcn_sched_barrier(0);
        }
      drain();  // tap NT1 (the first of conv2) has landed
      VFX_PC_BAR();
      // ---- P1: h = LeakyReLU(conv1 + b1) as fp16, zero outside the sequence (layout: resblock_act.hip) ----
      {
        f32x4 b1v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b1v[j] = *(const VFX_GLOBAL f32x4*)(p.b1 + wn * 32 + 8 * j + 4 * lh);
        bool f16_sat = false;
#pragma unroll
        for (int a = 0; a < WM; ++a) {
          const int m = a * 32 + l31;
          char* rowp = hs + m * HROW + ((wn >> 1) ^ (m & 1)) * CROW + 8 * lh;
          const int key = (m >> 1) & 7;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            f32x4 u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float tt = acc[a][4 * j + e] + b1v[j][e];
              u[e] = hval[a] ? fmaxf(tt, tt * slope) : 0.f;
              acc[a][4 * j + e] = 0.f;
            }
            *reinterpret_cast<uint2*>(rowp + ((((wn & 1) * 4 + j) ^ key) << 4)) =
                make_uint2(pack_f16x2(u[0], u[1], f16_sat), pack_f16x2(u[2], u[3], f16_sat));
          }
        }
        report_f16_saturation(f16_sat, p.flags);
      }
      VFX_PC_BAR();
      // ---- P2: conv2 from the resident h ----
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int g = NT1 + 3 * c + k;
          fetch(g + 1);  // past the end: the last tap again, never consumed
          wait_tap(g);
          int rows[WM];
          int lrow = l31;
          asm volatile("" : "+v"(lrow));  // as in conv1
#pragma unroll
          for (int a = 0; a < WM; ++a) {
            const int r = a * 32 + lrow + k - 1;
            rows[a] = r < 0 ? 0 : (r > MT - 1 ? MT - 1 : r);  // clamped rows only feed outputs that are masked anyway
          }
          mma(ring(g), hs, HROW, rows, c);
          __builtin_amdgcn_sched_barrier(0);
        }
      drain();
      fetch(0);  // the first tap of the next tile's conv1 travels during the epilogue passes
      VFX_PC_BAR();
      // ---- P3 .. P6: the four 64-channel passes, alternating staging buffers ----
#pragma unroll
      for (int pass = 0; pass < NEP; ++pass) {
        if ((wn >> 1) == pass) {
          stage(hs + (pass & 1) * STG_BYTES);
#pragma unroll
          for (int a = 0; a < WM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        }
        VFX_PC_BAR();
      }
    }
    drain();  // nothing of this wave may be in flight at the end
  }
}

bool resblock_pc_enabled() {
  static const bool on = getenv("VFX_RB_PC") && atoi(getenv("VFX_RB_PC")) != 0;
  return on;
}

void launch_resblock_pc(const ResBlockParams& hp, const ResBlockParams* dparams, hipStream_t stream) {
  VFX_CHECK(hp.asrc && hp.hionly && hp.C == 256 && hp.xa && hp.tile_m == 128, "resblock_pc: needs the 16-bit mode, C = 256, 128-position tiles");
  const int64_t ntiles = (int64_t)hp.B * hp.tiles_h * hp.tiles_w;
  VFX_CHECK(ntiles > 0 && ntiles < ((int64_t)1 << 30), "resblock_pc: bad tile count");
  const int cus = cu_count_of_current_device();
  const int per_block = (int)((ntiles + cus - 1) / cus);
  const int grid = (int)((ntiles + per_block - 1) / per_block);  // one block per CU (the block owns the CU's LDS); every block has >= 1 tile
  const size_t lds = (size_t)(256 / 64) * CPATCH + (size_t)128 * 256 * 2;  // 96 KB + 64 KB
  static uint64_t attr_devices = 0;
  if (first_use_on_current_device(attr_devices)) {
    VFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_resblock_pc<256>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
  }
  hipLaunchKernelGGL((k_resblock_pc<256>), dim3(grid), dim3(768), lds, stream, dparams, (int)ntiles, per_block);
  VFX_HIP(hipGetLastError());
}

}  // namespace vfx
