// resblock_act.hip -- one fused TFGAN ResStack layer of the 16-bit mode for the WIDE stacks (C = 256), whose trunk
// travels in two forms (vocoder.cpp): raw fp32 `x` (the residual stream) and activated fp16 `xa` = LeakyReLU(x), the
// operand of conv1 (2 bytes per element, vfx_internal.h / TapConvParams::out_act):
//
//     y  = x + conv2(LeakyReLU(conv1(xa) + b1)) + b2          conv1: k3, dilation d;  conv2: k3, dilation 1
//     ya = fp16(LeakyReLU_next(y))                             for the next consumer (next layer / upsampler)
//
// As two k_conv launches a layer moves 3.2 GB through HBM at C = 256 (xa, h written and re-read, the residual, y in
// both forms) and is bandwidth-bound on the second one; fused, h (128 positions x 256 channels of fp16 = 64 KB) never
// leaves the CU.
//
// One block = 8 waves = ALL 256 output channels of a tile of up to 128 h positions (wave = 128 positions x 32 channels,
// 64 accumulator registers): a weight fragment fetched from L2 feeds four MFMAs, and the A fragments of the tile are
// shared by the eight waves through LDS.  One block per CU (98 KB of LDS, 8 waves); its phases:
//   0. the whole xa patch of the tile -- C / 64 = 4 chunks of 64 channels, (128 + 2d) positions x 128 bytes each -- is
//      requested by LDS-DMA at once (no arithmetic on it: the producer applied the LeakyReLU; zero fill outside the
//      sequence by the buffer bound; pre-swizzled source pieces, cf. conv.hip).  ONE HBM round trip per tile; a weight
//      wait would retire the requests anyway (vmcnt counts in order), so nothing is gained by spreading them;
//   1. conv1: 4 chunks x 3 taps x 4 K = 16 steps on fp16 fragments (pack_conv mode 3), weights global -> VGPR ring with
//      hand-counted vmcnt as in k_conv / k_resblock;
//   2. h = LeakyReLU(acc + b1) as fp16, 512-byte rows, over the (dead) patch buffers;
//   3. conv2 from the resident h;
//   4. epilogue in two passes of 128 channels: + b2 + x (the raw residual: pass 0's is requested behind the last weight
//      fetch of conv2, pass 1's while pass 0 is added and stored), raw fp32 y and fp16 ya.
// Geometry (plan_resblock) as k_resblock: 1-D tiles for d <= 32, folded rows of d samples with vertical taps above.
#include "conv_common.h"
#include "vfx_internal.h"

namespace vfx {

// Timing-only ablation builds (-DVFX_RBA_ABL=mask, wrong results; scripts/abl_resblock_act.sh): 1 no weight loads,
// 2 no A-fragment reads from LDS, 4 no MFMAs, 8 no epilogue, 16 no patch / residual requests, 32 no h write.
#ifndef VFX_RBA_ABL
#define VFX_RBA_ABL 0
#endif

// MT = h positions per tile: 128 (one block per CU: 98 KB of LDS, 64 accumulator registers) or 64 (TWO blocks per CU: 64 KB,
// <= 128 VGPRs).  The timing-only ablation of the MT = 128 form (scripts/abl_resblock_act.sh) shows a layer that is the SUM of
// its memory phases (patch + residual in, y + ya out: 0.5 ms with all arithmetic removed) and its arithmetic (0.5 ms), because
// one block per CU cannot overlap the two and a wave cannot either: vmcnt retires loads AND stores in order, so the first
// weight wait of a later phase waits for everything issued before it.  Two co-resident blocks overlap each other's phases.
template <int C, int NW, int MT>
__global__ __launch_bounds__(NW * 64, MT == 128 ? 2 : 4) void k_resblock_act(const ResBlockParams* __restrict__ pp) {
  constexpr int NTHR = NW * 64;
  constexpr int KC = 64;                     // channels per chunk: a 128-byte row of fp16
  constexpr int NCH = C / KC;
  constexpr int PR = MT + 64;                // patch rows per chunk buffer (MT + 2 d for d <= 32; folded tiles need less)
  constexpr int PBYTES = PR * CROW;
  constexpr int RG = NTHR / 8;               // patch rows per DMA instruction group (8 lanes per row)
  constexpr int NG = PR / RG;                // DMA instructions per wave and chunk
  static_assert(PR % RG == 0, "patch rows must split into whole DMA groups");
  constexpr int WAVES_N = C / 32, WAVES_M = NW / WAVES_N, WM = MT / 32 / WAVES_M;
  static_assert(NW % WAVES_N == 0 && WAVES_M == 1 && WM * 32 == MT, "one wave = MT positions x 32 channels");
  constexpr bool PREFETCH_RES = false;       // the residual requested with the patch: measured -2 % at MT = 128, 64 more registers
  constexpr int WL = 4;                      // weight loads per tap and wave (four K = 16 fragments)
#ifndef VFX_RBA_RING
#define VFX_RBA_RING 2
#endif
  // Weight taps in flight: a tap is requested AHEAD taps before its use.  One tap of the MT = 128 form is 16 MFMAs (0.2 us) per
  // wave against an L2 round trip of ~0.6 us, but deeper rings measured no gain (-DVFX_RBA_RING=3 / 4: 0.974 / 0.999 ms median
  // per layer against 0.966 with one tap ahead, same box; 221 / 238 registers): the arithmetic phase does not wait for weights.
  // (Every MFMA takes its own 1 KB A fragment from LDS -- 128 B/clk per CU at full MFMA rate, half of what ds_read_b128 delivers --
  // and the two waves of a SIMD wait for those reads and for the block barriers between the phases.)
  constexpr int RING = MT == 128 ? VFX_RBA_RING : 2, AHEAD = RING - 1;
#ifndef VFX_RBA_PFNEXT
#define VFX_RBA_PFNEXT 1
#endif
  // The residual of epilogue pass p + 1 is requested while pass p is added and stored, and the one of pass 0 behind the LAST
  // weight fetch of conv2 (nothing younger is waited for before the epilogue): the four passes were four memory round trips
  // (the epilogue is 0.28 of the layer's 0.975 ms in the ablation).  One more set of residual registers.
  constexpr bool PFNEXT = VFX_RBA_PFNEXT && MT == 128 && !PREFETCH_RES && AHEAD == 1 && VFX_RBA_ABL == 0;
  constexpr int HROW = C * 2;                // bytes per h row (fp16)
  constexpr int NT1 = 3 * NCH;               // taps of conv1 (chunk-major); conv2 has as many
#ifndef VFX_RBA_EPC
#define VFX_RBA_EPC 128
#endif
  // Two passes of 128 channels, not four of 64: a pass is a memory round trip (its residual) and two block barriers.  Measured
  // per layer (median of 24 launches, same box): 64 / no prefetch 1.011 ms, 64 / next-pass prefetch 0.99, 128 / none 0.995,
  // 128 / prefetch 0.97 (250 registers).
  constexpr int EPC = MT == 128 ? VFX_RBA_EPC : 64, NEP = C / EPC;  // epilogue: passes of EPC channels (EPC / 32 waves stage a pass)
  constexpr int LDO = EPC + 4;               // staged output row (floats)

  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* const lds = reinterpret_cast<char*>(smem);

  const ResBlockParams& p = *pp;
  const int tid = threadIdx.x;
  int tile;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int tj = tile % p.tiles_w;
  const int ti = (tile / p.tiles_w) % p.tiles_h;
  const int img = tile / (p.tiles_w * p.tiles_h);
  const int T = p.T, d = p.dil, W1 = p.W1, TH = p.TH, PW = p.PW, P = p.P;
  const int rowstride = p.fold ? d : 0;
  const int j0 = tj * p.TWo;
  const int base_h = p.fold ? ti * TH * d + j0 - 1 : j0 - 1;  // position of h pixel (0, 0)
  const int base_x = base_h - d;                                // position of patch pixel (0, 0)
  const float slope = p.slope;

  const int lr = tid >> 3, cg = tid & 7;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int wn = wave_u % WAVES_N;
  const int l31 = lane & 31, lh = lane >> 5;
  const int key_l = (lr >> 1) & 7;  // swizzle key of patch rows lr + RG * q (RG / 2 is a multiple of 8)
  static_assert((RG / 2) % 8 == 0, "the key must not depend on the row group");

  // ---- the xa patch: every chunk at once -----------------------------------------------------------------------
  {
    unsigned voff[NG];
    unsigned okmask = 0;
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      const int prow = lr + RG * q;
      const int pi = prow / PW, pj = prow - pi * PW;
      const int pos = base_x + pi * rowstride + pj;
      const bool ok = (prow < P) & ((unsigned)pos < (unsigned)T);
      voff[q] = (unsigned)(img * T + pos) * (unsigned)(C * 2);
      okmask |= ok ? (1u << q) : 0u;
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(p.xa + c * kKC), 0, (int)(unsigned)((int64_t)p.B * T * C * 2 - (int64_t)c * kKC * 4), 0x00020000);
#pragma unroll
      for (int q = 0; q < NG; ++q) {
        // the lane's 16 bytes land in slot cg of its row: fetch source piece cg ^ key (piece p sits at slot p ^ key)
        const unsigned o = (okmask & (1u << q)) ? voff[q] + ((unsigned)(cg ^ key_l) << 4) : 0xfffffff0u;
        VFX_LDS void* l = (VFX_LDS void*)(lds + c * PBYTES + (RG * q + 8 * wave_u) * CROW);
        if (!(VFX_RBA_ABL & 16)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, (int)o, 0, 0, 0);
      }
    }
  }

  // ---- the residual: this thread's share of the raw x tile in epilogue layout (row r0 + q * RPP, 4 channels of every
  // 64-channel pass), requested NOW, behind the patch: it lands with it (one HBM round trip per tile, nothing is
  // waited for in the epilogue) at the price of 64 registers -- the block has the register file of a whole CU.
  constexpr int V = EPC / 4, RPP = NTHR / V, NPASS = MT / RPP;
  const int c4 = tid % V, r0 = tid / V;
  int opix[NPASS];
  f32x4 res[NEP][NPASS];
  f32x4 b2v[NEP];  // conv2's bias for this thread's channels of every pass: fetched here, not in the pass (an L2 round trip each)
#pragma unroll
  for (int pass = 0; pass < NEP; ++pass) b2v[pass] = *(const VFX_GLOBAL f32x4*)(p.b2 + pass * EPC + 4 * c4);
#pragma unroll
  for (int q = 0; q < NPASS; ++q) {
    const int m = r0 + q * RPP;  // h pixel of the staged row
    const int li = m / W1, lj = m - li * W1;
    const int pos = base_h + li * rowstride + lj;
    const bool ok = (li < TH) & (lj >= 1) & (lj <= W1 - 2) & ((unsigned)pos < (unsigned)T) & (!p.fold | (j0 + lj - 1 < d));
    opix[q] = ok ? img * T + pos : -1;
  }
  if constexpr (PREFETCH_RES) {
#pragma unroll
    for (int pass = 0; pass < NEP; ++pass)
#pragma unroll
      for (int q = 0; q < NPASS; ++q)
        res[pass][q] = (VFX_RBA_ABL & 16) ? f32x4{0.f, 0.f, 0.f, 0.f}
                                          : *(const VFX_GLOBAL f32x4*)(p.x + (int64_t)(opix[q] < 0 ? 0 : opix[q]) * C + pass * EPC + 4 * c4);
  }

  int arow1[WM], arow2[WM];  // A row of this lane's h pixel: in the patch (tap offset added) / in the h buffer
  bool hval[WM];             // that h pixel lies inside the tile's h grid and inside the sequence
#pragma unroll
  for (int a = 0; a < WM; ++a) {
    const int ml = a * 32 + l31;
    const int li = ml / W1, lj = ml - li * W1;
    arow1[a] = li < TH ? li * PW + lj : 0;
    arow2[a] = ml;
    const int pos = base_h + li * rowstride + lj;
    hval[a] = (li < TH) & ((unsigned)pos < (unsigned)T);
  }
  const unsigned nb_off = (unsigned)(wn * 1024 + lane * 4) * 4u;
  const int64_t ts = (int64_t)C * kKC;  // floats per (64-channel chunk, tap) of a weight tensor: C / 32 cout blocks of 1024

  f32x16 acc[WM];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  // ---- MFMA step: one tap of one 64-channel chunk; A rows `row[a]` of an LDS image with `stride` bytes per row ------
  // `chunk` >= 0: the image is h: chunk c of row r sits at chunk position c ^ (r & 1) -- the 512-byte rows all start on the
  // same half of the LDS banks and rows r, r + 1 share their swizzle key, so without the swap the reads of conv2 are
  // 2-way bank conflicts (cf. resblock.hip).  Patch rows (128-byte stride) alternate halves by themselves: chunk = -1.
  auto mma = [&](const BFrag& R, const char* img_base, int stride, const int (&row)[WM], int chunk) __attribute__((always_inline)) {
    const char* base[WM];
    int key[WM];
#pragma unroll
    for (int a = 0; a < WM; ++a) {
      base[a] = img_base + row[a] * stride;
      if (chunk >= 0) base[a] += (chunk ^ (row[a] & 1)) * CROW;
      key[a] = swz_key(row[a]);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const f16x8 bh = __builtin_bit_cast(f16x8, R.f[s]);
      f16x8 ah[WM];
#pragma unroll
      for (int a = 0; a < WM; ++a)
        ah[a] = (VFX_RBA_ABL & 2) ? __builtin_bit_cast(f16x8, R.f[(s + a) & 3])
                                  : *reinterpret_cast<const f16x8*>(base[a] + ((32 * s + 16 * lh) ^ key[a]));
      // D = W (A operand: rows = couts) x image rows (B operand: columns = pixels): lane = pixel, registers = four runs of
      // 4 consecutive couts
#pragma unroll
      for (int a = 0; a < WM; ++a) {
        if (VFX_RBA_ABL & 4) {
          asm volatile("" : "+v"(acc[a]) : "v"(bh), "v"(ah[a]));  // keeps the operands live
        } else {
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah[a], acc[a], 0, 0, 0);
        }
      }
    }
  };

  // ---- weight ring: global tap g (conv1: 0 .. NT1-1, conv2: NT1 .. 2*NT1-1) in register group g % RING -------------
  BFrag R0 = {}, R1 = {}, R2 = {}, R3 = {};
  auto ring = [&](int g) __attribute__((always_inline)) -> BFrag& {
    const int i = g % RING;
    return i == 0 ? R0 : (i == 1 ? R1 : (i == 2 ? R2 : R3));
  };
  auto fetch = [&](int g) __attribute__((always_inline)) {
    const float* w = g < NT1 ? p.w1 + g * ts : (g < 2 * NT1 ? p.w2 + (g - NT1) * ts : p.w2 + (NT1 - 1) * ts);
    if (!(VFX_RBA_ABL & 1) || g < AHEAD) load_b_asm(ring(g), w, nb_off);
  };
  // tap g's weights are older than the AHEAD fetches (WL loads each) issued after them
  auto wait_tap = [&](int g) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(WL * AHEAD) : "memory");
    use_b(ring(g));
  };
  auto drain = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    use_b(R0);
    use_b(R1);
    if constexpr (RING > 2) use_b(R2);
    if constexpr (RING > 3) use_b(R3);
  };

#pragma unroll
  for (int g = 0; g < AHEAD; ++g) fetch(g);
  drain();          // the patch and the first taps have landed (this wave's share)
  __syncthreads();  // ... and everybody else's

  // ---- phase 1: conv1 ---------------------------------------------------------------------------------------------
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int g = 3 * c + k;
      fetch(g + AHEAD);
      wait_tap(g);
      int rows[WM];
#pragma unroll
      for (int a = 0; a < WM; ++a) rows[a] = arow1[a] + p.poff[k];
      mma(ring(g), lds + c * PBYTES, CROW, rows, -1);
      __builtin_amdgcn_sched_barrier(0);
    }
  drain();          // the first taps of conv2 have landed
  __syncthreads();  // every wave is done reading the patch buffers that h overlays

  if (!(VFX_RBA_ABL & 32))
  // ---- phase 2: h = LeakyReLU(conv1 + b1) as fp16, zero outside the sequence -------------------------------------------
  // Lane (l31, lh) of M block a holds h pixel m = a*32 + l31 and, in registers 4j .. 4j+3, channels wn*32 + 8j + 4lh .. +3:
  // chunk wn >> 1 of the pixel's row, piece (wn & 1)*4 + j, half lh.
  {
    f32x4 b1v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b1v[j] = *(const VFX_GLOBAL f32x4*)(p.b1 + wn * 32 + 8 * j + 4 * lh);
    bool f16_sat = false;
#pragma unroll
    for (int a = 0; a < WM; ++a) {
      const int m = a * 32 + l31;
      char* rowp = lds + m * HROW + ((wn >> 1) ^ (m & 1)) * CROW + 8 * lh;  // chunk parity swap: see mma()
      const int key = (m >> 1) & 7;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = acc[a][4 * j + e] + b1v[j][e];
          u[e] = hval[a] ? fmaxf(t, t * slope) : 0.f;
          acc[a][4 * j + e] = 0.f;
        }
        *reinterpret_cast<uint2*>(rowp + ((((wn & 1) * 4 + j) ^ key) << 4)) =
            make_uint2(pack_f16x2(u[0], u[1], f16_sat), pack_f16x2(u[2], u[3], f16_sat));
      }
    }
    report_f16_saturation(f16_sat, p.flags);
  }
  __syncthreads();  // h is complete

  // ---- phase 3: conv2 from the resident h ------------------------------------------------------------------------------
  auto request_res = [&](int pass) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NPASS; ++q)
      res[pass][q] = *(const VFX_GLOBAL f32x4*)(p.x + (int64_t)(opix[q] < 0 ? 0 : opix[q]) * C + pass * EPC + 4 * c4);
  };
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int g = NT1 + 3 * c + k;
      if (PFNEXT && g >= 2 * NT1 - 2) {
        // the last two taps: the final (real) weight fetch, then the residual of epilogue pass 0 -- NPASS loads that are
        // younger than every weight still waited for
        if (g == 2 * NT1 - 2) {
          fetch(g + 1);
          request_res(0);
          asm volatile("s_waitcnt vmcnt(%0)" : : "n"(WL + NPASS) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NPASS) : "memory");
        }
        use_b(ring(g));
      } else {
        fetch(g + AHEAD);  // past the end: the last tap again, never consumed
        wait_tap(g);
      }
      int rows[WM];
#pragma unroll
      for (int a = 0; a < WM; ++a) {
        const int r = arow2[a] + k - 1;
        rows[a] = r < 0 ? 0 : (r > MT - 1 ? MT - 1 : r);  // clamped rows only feed outputs that are masked anyway
      }
      mma(ring(g), lds, HROW, rows, c);
      __builtin_amdgcn_sched_barrier(0);
    }
  if (!PFNEXT) drain();
  __syncthreads();  // every wave is done with h

  // ---- phase 4: y = conv2 + b2 + x, raw fp32 and (optionally) activated fp16 ---------------------------------------------
  const bool even = (tid & 1) == 0;
  const float aslope = p.act_slope;
  if (VFX_RBA_ABL & 8) {
    float keep = 0.f;
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) keep += acc[a][r];
    if (keep == 12345.678f) p.y[tid] = keep + (PREFETCH_RES ? res[0][0][0] + res[NEP - 1][NPASS - 1][3] : 0.f);  // keeps them live
    return;
  }
#pragma unroll
  for (int pass = 0; pass < NEP; ++pass) {
    if (wn / (EPC / 32) == pass) {
#pragma unroll
      for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = a * 32 + l31;
          *reinterpret_cast<f32x4*>(smem + row * LDO + (wn % (EPC / 32)) * 32 + 8 * j + 4 * lh) =
              f32x4{acc[a][4 * j], acc[a][4 * j + 1], acc[a][4 * j + 2], acc[a][4 * j + 3]};
        }
    }
    __syncthreads();  // the pass is staged
    const int ncol = pass * EPC + 4 * c4;
    const f32x4 bv = b2v[pass];
    f32x4 val[NPASS];
    if constexpr (PFNEXT) {
      if (pass + 1 < NEP) request_res(pass + 1);  // in flight while this pass is added and stored
    } else if constexpr (!PREFETCH_RES) {
      request_res(pass);
    }
#pragma unroll
    for (int q = 0; q < NPASS; ++q)
      val[q] = *reinterpret_cast<const f32x4*>(smem + (r0 + q * RPP) * LDO + 4 * c4) + bv + res[pass][q];
#pragma unroll
    for (int q = 0; q < NPASS; ++q)
      if (opix[q] >= 0) *(VFX_GLOBAL f32x4*)(p.y + (int64_t)opix[q] * C + ncol) = val[q];
    if (p.ya) {
      bool f16_sat = false;
#pragma unroll
      for (int q = 0; q < NPASS; ++q) {
        f32x4 u;
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = fmaxf(val[q][e], val[q][e] * aslope);
        const unsigned h01 = pack_f16x2(u[0], u[1], f16_sat), h23 = pack_f16x2(u[2], u[3], f16_sat);
        // quad_perm [1,0,3,2]: the even lane of a pair collects the pair's 8 consecutive channels (16 bytes)
        const unsigned g0 = (unsigned)__builtin_amdgcn_mov_dpp((int)h01, 0xB1, 0xf, 0xf, false);
        const unsigned g1 = (unsigned)__builtin_amdgcn_mov_dpp((int)h23, 0xB1, 0xf, 0xf, false);
        const u32x4 w = {h01, h23, g0, g1};
        if (opix[q] >= 0 && even)
          *(VFX_GLOBAL f32x4*)(p.ya + (int64_t)opix[q] * (C / 2) + (ncol >> 1)) = __builtin_bit_cast(f32x4, w);
      }
      report_f16_saturation(f16_sat, p.flags);
    }
    __syncthreads();  // the staged pass has been consumed
  }
}

static size_t resblock_act_lds_bytes(int C, int MT) {
  const size_t patches = (size_t)(C / 64) * (MT + 64) * CROW;
  const size_t h = (size_t)MT * C * 2;
  const size_t epi = (size_t)MT * ((MT == 128 ? VFX_RBA_EPC : 64) + 4) * 4;
  return std::max(std::max(patches, h), epi);
}

bool resblock_act_supported(int C) { return C == 256; }

// h positions per tile of the fused wide layer.  (A 64-position form with two blocks per CU was 34 % slower in round 3 --
// twice the weight traffic per result, a register budget that spilled -- and is gone; resblock_w64.hip is the two-block form.)
int resblock_act_tile() { return 128; }

template <int MT>
static void launch_rba(int grid, hipStream_t stream, const ResBlockParams* dparams) {
  const size_t lds = resblock_act_lds_bytes(256, MT);
  static uint64_t attr_devices = 0;
  if (first_use_on_current_device(attr_devices)) {
    VFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_resblock_act<256, 8, MT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
  }
  hipLaunchKernelGGL((k_resblock_act<256, 8, MT>), dim3(grid), dim3(512), lds, stream, dparams);
}

void launch_resblock_act(const ResBlockParams& hp, const ResBlockParams* dparams, hipStream_t stream) {
  VFX_CHECK(hp.asrc && hp.hionly && resblock_act_supported(hp.C) && hp.xa, "resblock_act: needs the 16-bit mode, C = 256 and the activated trunk");
  VFX_CHECK(hp.tile_m == 128, "resblock_act: tile of %d positions", hp.tile_m);
  const int64_t grid = (int64_t)hp.B * hp.tiles_h * hp.tiles_w;
  VFX_CHECK(grid > 0 && grid < ((int64_t)1 << 31), "resblock_act: bad grid");
  if (hp.tile_m == 128 && hp.patch_rows) return launch_resblock_w64(hp, dparams, stream);
  launch_rba<128>((int)grid, stream, dparams);
  VFX_HIP(hipGetLastError());
}

}  // namespace vfx
