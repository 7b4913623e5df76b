// resblock_rl.hip -- the fused ResStack layer of the 16-bit mode for C = 128 (the 14.7 kHz stack) with the x patch loaded
// THROUGH REGISTERS.
//
//     y = x + conv2(LeakyReLU(conv1(LeakyReLU(x)) + b1)) + b2      conv1: k3, dilation d;  conv2: k3, dilation 1
//
// k_resblock<128, 8> (resblock.hip) brings the raw fp32 patch in by LDS-DMA, one 32-channel chunk (24 KB) at a time through two
// buffers, and transforms it in place: conv1 of a tile is a chain of FOUR memory round trips with 0.2 us of MFMAs between them
// (26 us per tile for 1.3 us of arithmetic), and the residual is read a second time in the epilogue.  Here
//   * every thread fetches its share of the WHOLE patch at once (global_load, 12 x 16 bytes: 192 positions x 512 bytes over 512
//     threads; out-of-sequence rows are zeros) -- one round trip per tile;
//   * the patch goes to LDS as fp16 operands already (LeakyReLU applied; 256-byte rows = two 64-channel chunks, swizzled
//     pieces, chunk parity swap as for h): 48 KB for all 128 channels instead of 2 x 24 KB per 32;
//   * the centre rows stay in registers (32 VGPRs) and are the residual of the epilogue, whose thread -> (row, 4 channels) map
//     is the map of those loads: x is read ONCE;
//   * weights: fp16 fragments of 64-channel chunks (pack_conv mode 3), global -> VGPR ring with hand-counted vmcnt.  A wave
//     owns 32 output channels over ALL 128 positions (a fragment feeds four MFMAs) and keeps THREE taps in flight: 0.6 us of
//     MFMAs between a request and its use, an L2 round trip.  (With 64-position waves and one tap ahead -- the geometry of
//     k_resblock -- this kernel ran 1.22 ms per layer against 0.88: every tap exposed most of its L2 latency.)
// Block = 4 waves with 256 registers; LDS: patch 48 KB + h 32 KB = 80 KB (the staged accumulators overlay both): two blocks per CU.
// Tile geometry as k_resblock (plan_resblock): 128 h positions, 1-D for d <= 32, folded rows of d samples above.
#include "conv_common.h"
#include "vfx_internal.h"

namespace vfx {

template <int C>
__global__ __launch_bounds__(C * 2, 2) void k_resblock_rl(const ResBlockParams* __restrict__ pp) {
  static_assert(C == 128, "one wave per 32 output channels, all 128 positions: 4 waves");
  constexpr int NW = C / 32;                 // 4 waves along the channels; 256 registers each
  constexpr int NTHR = NW * 64;
  constexpr int MT = 128;                    // h positions per tile
  constexpr int PR = MT + 64;                // patch rows
  constexpr int NCH = C / 64;                // 64-channel chunks (128 bytes of fp16)
  constexpr int ROWB = C * 2;                // bytes per LDS row (patch and h)
  constexpr int LPR = C / 4;                 // lanes per raw row (16 bytes each)
  constexpr int RQ = NTHR / LPR;             // rows per load group (8)
  constexpr int NCQ = MT / RQ;               // centre loads per thread (16)
  constexpr int NHQ = 64 / RQ;               // halo loads per thread (8)
  constexpr int WM = 4;                      // wave = 128 positions x 32 channels: a weight fragment feeds four MFMAs
  constexpr int RING = 4, AHEAD = RING - 1;  // weight taps in flight: 3 x 16 MFMAs (0.6 us) ahead of their use -- an L2 round trip
  constexpr int WL = 4;                      // weight loads per tap and wave (four K = 16 fragments)
  constexpr int NT1 = 3 * NCH;               // taps of conv1 (chunk-major); conv2 has as many
  constexpr int H_OFF = PR * ROWB;           // h behind the patch
  constexpr int LDO = C + 4;                 // staged output row (floats)
  static_assert(MT * LDO * 4 <= (PR + MT) * ROWB, "staging must fit over patch + h");

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
  const int c0 = p.fold ? PW : d;  // patch row of h pixel 0: patch row m + c0 holds the input sample AT h pixel m (the residual)
  const float slope = p.slope;

  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int wm = 0, wn = wave_u;
  const int l31 = lane & 31, lh = lane >> 5;
  const int lr = tid / LPR, cg = tid % LPR;

  // ---- the raw patch: centre q = patch row lr + RQ q + c0 (= h pixel lr + RQ q), halo q = the rows in front of / behind it ----
  auto rel_of = [&](int pr) __attribute__((always_inline)) {  // sample offset from the patch origin, or -1: no such row
    const int pi = p.fold ? pr / PW : 0;
    return pr < P ? pi * rowstride + (pr - pi * PW) : -1;
  };
  auto hrow = [&](int q) __attribute__((always_inline)) {
    const int hr = lr + RQ * q;
    return hr < c0 ? hr : hr + MT;
  };
  f32x4 K[NCQ], PH[NHQ];
  {
    const float* xi = p.x + (int64_t)img * T * C + 4 * cg;
#pragma unroll
    for (int q = 0; q < NCQ; ++q) {
      const int rel = rel_of(lr + RQ * q + c0), pos = base_x + rel;
      K[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (rel >= 0 && (unsigned)pos < (unsigned)T) K[q] = *(const VFX_GLOBAL f32x4*)(xi + (int64_t)pos * C);
    }
#pragma unroll
    for (int q = 0; q < NHQ; ++q) {
      const int rel = rel_of(hrow(q)), pos = base_x + rel;
      PH[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (rel >= 0 && (unsigned)pos < (unsigned)T) PH[q] = *(const VFX_GLOBAL f32x4*)(xi + (int64_t)pos * C);
    }
  }

  int arow1[WM];  // patch row of this lane's h pixels (tap offset added per tap)
  bool hval[WM];  // that h pixel lies inside the tile's h grid and inside the sequence
#pragma unroll
  for (int a = 0; a < WM; ++a) {
    const int ml = (wm * WM + a) * 32 + l31;
    const int li = ml / W1, lj = ml - li * W1;
    arow1[a] = li < TH ? li * PW + lj : 0;
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

  // ---- MFMA step: one tap of one 64-channel chunk; A rows `row[a]` of an LDS image of ROWB-byte rows.  Chunk c of row r sits at
  // chunk position c ^ (r & 1): the rows all start on the same LDS bank and rows r, r + 1 share their swizzle key (resblock.hip).
  auto mma = [&](const BFrag& R, const char* img_base, const int (&row)[WM], int chunk) __attribute__((always_inline)) {
    const char* base[WM];
    int key[WM];
#pragma unroll
    for (int a = 0; a < WM; ++a) {
      base[a] = img_base + row[a] * ROWB + (chunk ^ (row[a] & 1)) * CROW;
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

  // ---- weight ring: global tap g (conv1: 0 .. NT1-1, conv2: NT1 .. 2*NT1-1) in register group g % RING; tap g + AHEAD is
  // requested right before tap g is consumed, so tap g is older than exactly AHEAD fetches (taps past the end: the last one again)
  BFrag R0 = {}, R1 = {}, R2 = {}, R3 = {};
  auto ring = [&](int g) __attribute__((always_inline)) -> BFrag& {
    return (g & 3) == 0 ? R0 : ((g & 3) == 1 ? R1 : ((g & 3) == 2 ? R2 : R3));
  };
  auto fetch = [&](int g) __attribute__((always_inline)) {
    const float* w = g < NT1 ? p.w1 + g * ts : (g < 2 * NT1 ? p.w2 + (g - NT1) * ts : p.w2 + (NT1 - 1) * ts);
    load_b_asm(ring(g), w, nb_off);
  };
  auto wait_tap = [&](int g) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(WL * AHEAD) : "memory");
    use_b(ring(g));
  };
  auto drain = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    use_b(R0);
    use_b(R1);
    use_b(R2);
    use_b(R3);
  };

  // ---- the landed patch: LeakyReLU, fp16, to LDS in operand form ------------------------------------------------------------
  // this thread's 4 channels 4 cg .. 4 cg + 3: chunk cg / 16, 16-byte piece (cg / 2) % 8, half cg % 2
  {
    bool sat = false;
    auto to_patch = [&](const f32x4& raw, int pr) __attribute__((always_inline)) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(raw[e], raw[e] * slope);
      *reinterpret_cast<uint2*>(lds + pr * ROWB + (((cg >> 4) ^ (pr & 1)) * CROW) + ((((cg >> 1) & 7) ^ ((pr >> 1) & 7)) << 4) + 8 * (cg & 1)) =
          make_uint2(pack_f16x2(v[0], v[1], sat), pack_f16x2(v[2], v[3], sat));
    };
#pragma unroll
    for (int q = 0; q < NCQ; ++q)
      if (lr + RQ * q + c0 < P) to_patch(K[q], lr + RQ * q + c0);
#pragma unroll
    for (int q = 0; q < NHQ; ++q)
      if (hrow(q) < P) to_patch(PH[q], hrow(q));
    report_f16_saturation(sat, p.flags);
  }
#pragma unroll
  for (int g = 0; g < AHEAD; ++g) fetch(g);
  __syncthreads();  // the patch is complete

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
      mma(ring(g), lds, rows, c);
      __builtin_amdgcn_sched_barrier(0);
    }
  // (the first AHEAD taps of conv2 are in flight: the compiler's own wait for b1 below retires them too, harmlessly)

  // ---- phase 2: h = LeakyReLU(conv1 + b1) as fp16, zero outside the sequence (layout as the patch) --------------------------
  // Lane (l31, lh) of M block a holds h pixel m and, in registers 4j .. 4j+3, channels wn*32 + 8j + 4lh .. +3: chunk wn >> 1
  // of the pixel's row, piece (wn & 1)*4 + j, half lh.
  {
    f32x4 b1v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b1v[j] = *(const VFX_GLOBAL f32x4*)(p.b1 + wn * 32 + 8 * j + 4 * lh);
    bool sat = false;
#pragma unroll
    for (int a = 0; a < WM; ++a) {
      const int m = (wm * WM + a) * 32 + l31;
      char* rowp = lds + H_OFF + m * ROWB + ((wn >> 1) ^ (m & 1)) * CROW + 8 * lh;
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
        *reinterpret_cast<uint2*>(rowp + ((((wn & 1) * 4 + j) ^ key) << 4)) = make_uint2(pack_f16x2(u[0], u[1], sat), pack_f16x2(u[2], u[3], sat));
      }
    }
    report_f16_saturation(sat, p.flags);
  }
  __syncthreads();  // h is complete

  // ---- phase 3: conv2 from the resident h ------------------------------------------------------------------------------
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int g = NT1 + 3 * c + k;
      fetch(g + AHEAD);  // past the end: the last tap again, never consumed
      wait_tap(g);
      int rows[WM];
#pragma unroll
      for (int a = 0; a < WM; ++a) {
        const int r = (wm * WM + a) * 32 + l31 + k - 1;
        rows[a] = r < 0 ? 0 : (r > MT - 1 ? MT - 1 : r);  // clamped rows only feed outputs that are masked anyway
      }
      mma(ring(g), lds + H_OFF, rows, c);
      __builtin_amdgcn_sched_barrier(0);
    }
  drain();
  __syncthreads();  // every wave is done with the patch and h: the staged tile overlays them

  // ---- phase 4: y = conv2 + x + b2 in the layout of the centre loads; optionally ya = fp16(LeakyReLU(y, act_slope)) --------
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (wm * WM + a) * 32 + l31;
      *reinterpret_cast<f32x4*>(smem + row * LDO + wn * 32 + 8 * j + 4 * lh) =
          f32x4{acc[a][4 * j], acc[a][4 * j + 1], acc[a][4 * j + 2], acc[a][4 * j + 3]};
    }
  __syncthreads();
  {
    float* yi = p.y + (int64_t)img * T * C + 4 * cg;
    const f32x4 b2v = *(const VFX_GLOBAL f32x4*)(p.b2 + 4 * cg);
    const bool even = (tid & 1) == 0;
    const float aslope = p.act_slope;
    bool sat = false;
#pragma unroll
    for (int q = 0; q < NCQ; ++q) {
      const int m = lr + RQ * q;
      const int li = m / W1, lj = m - li * W1;
      const int pos = base_h + li * rowstride + lj;
      const bool ok = (li < TH) & (lj >= 1) & (lj <= W1 - 2) & ((unsigned)pos < (unsigned)T) & (!p.fold | (j0 + lj - 1 < d));
      const f32x4 val = (*reinterpret_cast<const f32x4*>(smem + m * LDO + 4 * cg) + K[q]) + b2v;
      if (ok) *(VFX_GLOBAL f32x4*)(yi + (int64_t)pos * C) = val;
      if (p.ya) {
        f32x4 u;
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = fmaxf(val[e], val[e] * aslope);
        const unsigned h01 = pack_f16x2(u[0], u[1], sat), h23 = pack_f16x2(u[2], u[3], sat);
        // quad_perm [1,0,3,2]: the even lane of a pair collects the pair's 8 consecutive channels (16 bytes)
        const unsigned g0 = (unsigned)__builtin_amdgcn_mov_dpp((int)h01, 0xB1, 0xf, 0xf, false);
        const unsigned g1 = (unsigned)__builtin_amdgcn_mov_dpp((int)h23, 0xB1, 0xf, 0xf, false);
        const u32x4 w = {h01, h23, g0, g1};
        if (ok && even) *(VFX_GLOBAL f32x4*)(p.ya + ((int64_t)img * T + pos) * (C / 2) + 2 * cg) = __builtin_bit_cast(f32x4, w);
      }
    }
    if (p.ya) report_f16_saturation(sat, p.flags);
  }
}

// Opt-in (VFX_RB_RL=1): measured 0.91 ms per 1-D layer and 1.06 ms per folded one against 0.88 / 1.00 ms of k_resblock<128, 8>
// on gsr16x10 (profiles/r02_persistent_kernels_ab.json).  It moves 3.0 GB per layer instead of 3.65 GB (no second read of x),
// but two blocks of four waves per CU overlap less than two blocks of eight: k_resblock sits at the 4.2 TB/s that mixed
// read / write streams reach, this kernel at 3.3.
bool resblock_rl_enabled() {
  static const bool on = getenv("VFX_RB_RL") && atoi(getenv("VFX_RB_RL")) != 0;
  return on;
}

void launch_resblock_rl(const ResBlockParams& hp, const ResBlockParams* dparams, hipStream_t stream) {
  VFX_CHECK(hp.rl && hp.hionly && hp.C == 128 && !hp.geo2d && !hp.asrc && hp.tile_m == 128, "resblock_rl: needs the 16-bit mode and C = 128");
  const int64_t grid = (int64_t)hp.B * hp.tiles_h * hp.tiles_w;
  VFX_CHECK(grid > 0 && grid < ((int64_t)1 << 31), "resblock_rl: bad grid");
  const size_t lds = (size_t)(128 + 64 + 128) * 256;  // patch + h = 80 KB: two blocks (of 4 waves with 256 registers) per CU
  static uint64_t attr_devices = 0;
  if (first_use_on_current_device(attr_devices)) {
    VFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_resblock_rl<128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  hipLaunchKernelGGL((k_resblock_rl<128>), dim3((unsigned)grid), dim3(256), lds, stream, dparams);
  VFX_HIP(hipGetLastError());
}

}  // namespace vfx
