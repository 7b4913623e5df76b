// resblock_s256.hip -- one fused TFGAN ResStack layer of the 16-bit mode at C = 256 on a SINGLE-FORM trunk
//
//     y = x + conv2(LeakyReLU(conv1(LeakyReLU(x)) + b1)) + b2        conv1: k3, dilation d;  conv2: k3, dilation 1
//
// x and y are the raw fp32 trunk; nothing else crosses HBM (the last layer of the stack also writes the activated fp16 form the
// next upsampler reads).
//
// Opt-in with VFX_TUNE_WIDE_SINGLE_FORM -- see resblock_s256_enabled() below for what it measured.
//
// Why it was written (round-3 phase stamps and PMC): the two-form layer (resblock_w64.hip: x and an activated fp16 copy xa
// in, y and ya out) moves 400 KB per 128-position tile -- 12 bytes per element plus the halo -- and a CU gets ~10 B/clk of the
// fabric: 40 k of the 44.9 k cycles a CU has per tile.  That layer is traffic-bound on its own data layout.  The single-form
// layer needs the tile's raw centre rows on chip for the residual: at 128 positions that is 128 registers per thread beside 128
// accumulators.  Here a tile is 64 positions: a thread keeps 16 rows x 4 channels (64 registers), a wave owns 64 couts x 64
// positions (64 accumulator registers; a pixel fragment feeds two MFMAs, a weight fragment two), a block is 4 waves, its LDS --
// the fp16 operand patch as four 64-channel chunk buffers of 96 rows (48 KB), h (32 KB) and the staged tile (65 KB) overlaying
// one another -- fits twice into a CU.  Per 64 positions a block reads 64 + 2 (d + 1) rows of 1 KB once and writes 64: 8.3-9.3
// bytes per element.  The patch goes global -> registers -> operand rows (resblock_r128.hip), the convolutions are those of
// resblock_w64.hip (64-channel chunk rows, four K steps per tap, fragment reads software-pipelined one K step ahead, weight ring
// one tap ahead, no scheduling barriers and no memory clobbers in the compute phases).
//
// Tile geometry: plan_resblock with tile_m = 64, patch_rows = 96 (1-D tiles for d <= 16, folded rows of d samples with 4 x 16 h
// tiles above).  Weights: pack_conv mode 3 (fp16, 64-channel chunks), the packing of resblock_w64.hip.
#include "conv_common.h"
#include "vfx_internal.h"

namespace vfx {

namespace {
constexpr int S256_PR = 96;  // patch rows per chunk buffer
constexpr int S256_MT = 64;  // h positions per tile
}

__global__ __launch_bounds__(256, 2) void k_resblock_s256(const ResBlockParams* __restrict__ pp) {
  constexpr int C = 256, NW = 4, NTHR = NW * 64, MT = S256_MT;
  constexpr int NCH = C / 64;                // 64-channel chunks: 128-byte rows of fp16
  static_assert(NCH == NW, "one wave per 64 output channels");
  constexpr int PR = S256_PR;
  constexpr int PBYTES = PR * CROW;
  constexpr int WM = MT / 32;                // 32-position blocks per wave
  constexpr int WL = 8;                      // weight loads per tap and wave: 2 cout blocks x 4 K steps
  constexpr int HROW = C * 2;                // bytes per h row (fp16)
  constexpr int NT1 = 3 * NCH;               // taps of conv1 (chunk-major); conv2 has as many
  constexpr int LDO = C + 4;                 // staged output row (floats)
  constexpr int V = C / 4, RPP = NTHR / V;   // 64 float4 per row, 4 rows per step
  constexpr int KEEP = MT / RPP;             // centre rows per thread (16: rows rt + 4 j)
  constexpr int NHALO = (PR - MT) / RPP;     // halo rows per thread (8)
  constexpr int NROW = KEEP + NHALO;
  static_assert(V == 64 && RPP == 4 && KEEP == 16 && NHALO == 8, "geometry");

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
  VFX_TS_DECL;
  VFX_TS(0);
  // tile -> (image, tile row, tile column) with the host's reciprocals (plan_resblock)
  const int img = div_recip(tile, p.inv_tiles_per_img);
  const int trem = tile - img * (p.tiles_w * p.tiles_h);
  const int ti = div_recip(trem, p.inv_tiles_w);
  const int tj = trem - ti * p.tiles_w;
  const int T = p.T, d = p.dil, W1 = p.W1, TH = p.TH, PW = p.PW, P = p.P;
  const int rowstride = p.fold ? d : 0;
  const int j0 = tj * p.TWo;
  const int base_h = p.fold ? ti * TH * d + j0 - 1 : j0 - 1;  // position of h pixel (0, 0)
  const int base_x = base_h - d;                                // position of patch pixel (0, 0)
  const float slope = p.slope;
  const unsigned inv_pw = p.inv_pw, inv_w1 = p.inv_w1;          // m / W1 and prow / PW as multiply-shift (exact: rows < 512)

  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int l31 = lane & 31, lh = lane >> 5;

  int arow1[WM];   // A row of this lane's h pixel in the patch (tap offset to be added)
  bool hval[WM];   // that h pixel lies inside the tile's h grid and inside the sequence
#pragma unroll
  for (int a = 0; a < WM; ++a) {
    const int ml = a * 32 + l31;
    const int li = (int)(((unsigned)ml * inv_w1) >> 20), lj = ml - li * W1;
    arow1[a] = li < TH ? li * PW + lj : 0;
    const int pos = base_h + li * rowstride + lj;
    hval[a] = (li < TH) & ((unsigned)pos < (unsigned)T);
  }
  // weights: (64-channel chunk, tap) blocks of C / 32 cout blocks x 1024 floats; this wave's cout blocks are 2 w, 2 w + 1
  const unsigned nb_off = (unsigned)(2 * wave_u * 1024 + lane * 4) * 4u;
  const int64_t ts = (int64_t)C * kKC;

  f32x16 acc[2][WM];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[cb][a][r] = 0.f;

  // Everything the compute phases read from the parameter block, once: the waits below are `asm volatile` statements, and a
  // reload of p.poff / p.w1 behind one of them would be a scalar load + lgkmcnt(0) -- which also drains the LDS reads in flight.
  const float* const w1p = p.w1;
  const float* const w2p = p.w2;
  const int poff0 = p.poff[0], poff1 = p.poff[1], poff2 = p.poff[2];

  // ---- weight ring: global tap g (conv1: 0 .. NT1-1, conv2: NT1 .. 2*NT1-1) in register group g & 1, one tap ahead -----------
  // (no "memory" clobbers in the compute phases: an LDS fragment read may move across a weight fetch / wait)
  BFrag R0a = {}, R0b = {}, R1a = {}, R1b = {};  // [ring slot][cout block]
  auto load_w = [&](BFrag& R, const float* wtap) __attribute__((always_inline)) {
    asm volatile(
        "s_nop 4\n\t"
        "global_load_dwordx4 %0, %4, %5\n\t"
        "global_load_dwordx4 %1, %4, %5 offset:1024\n\t"
        "global_load_dwordx4 %2, %4, %5 offset:2048\n\t"
        "global_load_dwordx4 %3, %4, %5 offset:3072"
        : "=&v"(R.f[0]), "=&v"(R.f[1]), "=&v"(R.f[2]), "=&v"(R.f[3])
        : "v"(nb_off), "s"(wtap));
  };
  auto fetch = [&](int g) __attribute__((always_inline)) {
    const float* w = g < NT1 ? w1p + g * ts : w2p + (g - NT1) * ts;
    if (g & 1) {
      load_w(R1a, w);
      load_w(R1b, w + 1024);
    } else {
      load_w(R0a, w);
      load_w(R0b, w + 1024);
    }
  };
  // The registers of ring slot s are readable behind this statement (a counted s_waitcnt precedes it in program order: asm
  // volatile statements keep their order); every reader depends on its outputs, so no scheduling barrier is needed.
  auto use_slot = [&](int s) __attribute__((always_inline)) {
    if (s) {
      asm volatile("" : "+v"(R1a.f[0]), "+v"(R1a.f[1]), "+v"(R1a.f[2]), "+v"(R1a.f[3]), "+v"(R1b.f[0]), "+v"(R1b.f[1]), "+v"(R1b.f[2]),
                   "+v"(R1b.f[3]));
    } else {
      asm volatile("" : "+v"(R0a.f[0]), "+v"(R0a.f[1]), "+v"(R0a.f[2]), "+v"(R0a.f[3]), "+v"(R0b.f[0]), "+v"(R0b.f[1]), "+v"(R0b.f[2]),
                   "+v"(R0b.f[3]));
    }
  };

  fetch(0);  // issued BEFORE the patch loads: older than every load the compiler counts, in flight through the conversion

  // ---- the x patch: global -> registers -> fp16 operand rows in LDS; the 64 centre rows stay in registers as the residual --------
  // Thread (rt, c4) = (tid / 64, tid % 64) -- the thread that will STORE channels 4 c4 .. + 3 of the output rows m = rt + 4 j in the
  // epilogue -- loads exactly those rows of x (patch rows m + off, off = d for 1-D tiles, PW = one patch row up for folded ones:
  // the centre window of the patch) and keeps their raw values: the residual is added in the output pass from registers, and x is
  // read from memory ONCE.  The 32 halo rows around the window are shared out eight per thread.  A wave's load instruction is one
  // whole 1 KB row of x.  Operand form: channel c of a row lives in chunk c / 64, piece (c % 64) / 8, as fp16; the 8 bytes of
  // the thread's four channels go to slot (piece ^ key(row)), half c4 & 1, of the chunk's 128-byte row.  Rows outside the
  // sequence / the patch are loaded beyond the descriptor's bound: zeros, and LeakyReLU(0) = 0.
  const int off = p.fold ? PW : d;  // <= 32 (launch_resblock_s256)
  const int rt = tid >> 6, c4 = tid & 63;
  f32x4 keep[KEEP];
  {
    unsigned f16_sat = 0;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)(unsigned)((int64_t)p.B * T * C * 4), 0x00020000);
    const unsigned lane_off = (unsigned)(c4 * 16);
    char* const cbase = lds + (c4 >> 4) * PBYTES + 8 * (c4 & 1);
    const int piece = (c4 & 15) >> 1;
    int prow[NROW];
    u32x4 raw[NROW];
#pragma unroll
    for (int j = 0; j < NROW; ++j) {
      const int hh = rt + RPP * (j - KEEP);
      prow[j] = j < KEEP ? rt + RPP * j + off : (hh < off ? hh : hh + MT);
      const int pi = (int)(((unsigned)prow[j] * inv_pw) >> 20), pj = prow[j] - pi * PW;
      const int pos = base_x + pi * rowstride + pj;
      const bool ok = (prow[j] < P) & ((unsigned)pos < (unsigned)T);
      const unsigned o = ok ? (unsigned)(img * T + pos) * (unsigned)(C * 4) + lane_off : 0xfffffff0u;
      raw[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)o, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);  // all loads are in flight before the first row is converted
    VFX_TS(1);  // patch requested
#pragma unroll
    for (int j = 0; j < NROW; ++j) {
      const f32x4 r = __builtin_bit_cast(f32x4, raw[j]);
      if (j < KEEP) keep[j] = r;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(r[e], r[e] * slope);
      const int key = (prow[j] >> 1) & 7;
      *reinterpret_cast<uint2*>(cbase + prow[j] * CROW + ((piece ^ key) << 4)) =
          make_uint2(pack_f16x2(v[0], v[1], f16_sat), pack_f16x2(v[2], v[3], f16_sat));
    }
    report_f16_saturation(f16_sat_bits_bad(f16_sat), p.flags);
  }
  VFX_TS(2);  // loaded and converted
  VFX_TS(3);
  __syncthreads();  // the operand rows of every wave are visible
  VFX_TS(4);

  // ---- pixel fragments: software-pipelined one K step (4 MFMAs = 128 cycles) ahead ---------------------------------------------------
  // rb / kx: byte offset of the lane's row of position block a in the LDS image of tap g, and its swizzle key xor the lane's
  // half (recomputed per tap behind an opaque statement)
  int rb[2][WM], kx[2][WM];
  auto prep1 = [&](int g) __attribute__((always_inline)) {  // conv1: the patch chunk of tap g, rows arow1 + tap offset
    const int c = g / 3, k = g % 3;
#pragma unroll
    for (int a = 0; a < WM; ++a) {
      int r = arow1[a];
      asm volatile("" : "+v"(r));
      const int row = r + (k == 0 ? poff0 : (k == 1 ? poff1 : poff2));
      rb[g & 1][a] = c * PBYTES + row * CROW;
      kx[g & 1][a] = swz_key(row) ^ (16 * lh);
    }
  };
  auto prep2 = [&](int g) __attribute__((always_inline)) {  // conv2: h rows m + k - 1; chunk c of row r sits at chunk position c ^ (r & 1)
    const int c = (g - NT1) / 3, k = (g - NT1) % 3;
    int lrow = l31;
    asm volatile("" : "+v"(lrow));
#pragma unroll
    for (int a = 0; a < WM; ++a) {
      const int r0 = a * 32 + lrow + k - 1;
      const int row = r0 < 0 ? 0 : (r0 > MT - 1 ? MT - 1 : r0);  // clamped rows only feed outputs that are masked anyway
      rb[g & 1][a] = row * HROW + (c ^ (row & 1)) * CROW;
      kx[g & 1][a] = swz_key(row) ^ (16 * lh);
    }
  };
  f16x8 pxE[WM], pxO[WM];
  auto rd = [&](f16x8 (&px)[WM], int g, int st) __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < WM; ++a) px[a] = *reinterpret_cast<const f16x8*>(lds + rb[g & 1][a] + (kx[g & 1][a] ^ (32 * st)));
  };
  // one K step of tap g on ring slot g & 1: D = W (A operand: rows = couts) x image rows (B operand: columns = pixels): lane =
  // pixel, registers = four runs of 4 consecutive couts
  auto mm = [&](const f16x8 (&px)[WM], int g, int st) __attribute__((always_inline)) {
    const BFrag& Ra = (g & 1) ? R1a : R0a;
    const BFrag& Rb = (g & 1) ? R1b : R0b;
    const f16x8 wa = __builtin_bit_cast(f16x8, Ra.f[st]);
    const f16x8 wb = __builtin_bit_cast(f16x8, Rb.f[st]);
#pragma unroll
    for (int a = 0; a < WM; ++a) {
      acc[0][a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa, px[a], acc[0][a], 0, 0, 0);
      acc[1][a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb, px[a], acc[1][a], 0, 0, 0);
    }
  };
  // taps g0 .. g1-1 of one convolution (prep = prep1 / prep2); tap g1 (if it exists in this launch) is fetched but not used here
  auto conv = [&](auto prep, int g0, int g1, bool fetch_past) __attribute__((always_inline)) {
    prep(g0);
    rd(pxE, g0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, WM, 0);  // the prologue's reads are a group of their own: the pattern below starts behind them
#pragma unroll
    for (int g = g0; g < g1; ++g) {
      if (g + 1 < g1 || fetch_past) {
        fetch(g + 1);
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"(WL));
      } else {
        asm volatile("s_waitcnt vmcnt(0)");
      }
      use_slot(g & 1);
      if (g + 1 < g1) prep(g + 1);
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const bool last = g + 1 == g1 && st == 3;
        if (st & 1) {
          if (!last) { if (st == 3) rd(pxE, g + 1, 0); else rd(pxE, g, st + 1); }
          mm(pxO, g, st);
        } else {
          rd(pxO, g, st + 1);
          mm(pxE, g, st);
        }
#pragma unroll
        for (int i = 0; i < WM; ++i) {  // 1 fragment read, then 2 MFMAs
          if (!last) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
      }
    }
  };

  // ---- phase 1: conv1 (its last tap fetches the first tap of conv2) -------------------------------------------------------------
  conv(prep1, 0, NT1, true);
  VFX_TS(5);
  __syncthreads();  // every wave is done reading the patch buffers that h overlays
  VFX_TS(6);

  // ---- phase 2: h = LeakyReLU(conv1 + b1) as fp16, zero outside the sequence ---------------------------------------------------
  // Lane (l31, lh) of position block a holds h pixel m = a*32 + l31 and, in registers 4j .. 4j+3 of cout block cb, channels
  // (2w + cb)*32 + 8j + 4lh .. +3: chunk w of the pixel's row, piece cb*4 + j, half lh.
  {
    unsigned f16_sat = 0;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      f32x4 b1v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b1v[j] = *(const VFX_GLOBAL f32x4*)(p.b1 + (2 * wave_u + cb) * 32 + 8 * j + 4 * lh);
#pragma unroll
      for (int a = 0; a < WM; ++a) {
        const int m = a * 32 + l31;
        char* rowp = lds + m * HROW + (wave_u ^ (m & 1)) * CROW + 8 * lh;  // chunk parity swap: see prep2()
        const int key = (m >> 1) & 7;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4 u;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float t = acc[cb][a][4 * j + e] + b1v[j][e];
            u[e] = hval[a] ? fmaxf(t, t * slope) : 0.f;
            acc[cb][a][4 * j + e] = 0.f;
          }
          *reinterpret_cast<uint2*>(rowp + (((cb * 4 + j) ^ key) << 4)) =
              make_uint2(pack_f16x2(u[0], u[1], f16_sat), pack_f16x2(u[2], u[3], f16_sat));
        }
      }
    }
    report_f16_saturation(f16_sat_bits_bad(f16_sat), p.flags);
  }
  VFX_TS(7);
  __syncthreads();  // h is complete
  VFX_TS(8);

  // ---- phase 3: conv2 from the resident h ----------------------------------------------------------------------------------------
  conv(prep2, NT1, 2 * NT1, false);
  VFX_TS(9);
  __syncthreads();  // every wave is done with h
  VFX_TS(10);

  // ---- phase 4: y = conv2 + b2 + x: the accumulators staged in LDS, whole rows read back, the kept x rows added, stored ----------
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = a * 32 + l31;
        *reinterpret_cast<f32x4*>(smem + row * LDO + (2 * wave_u + cb) * 32 + 8 * j + 4 * lh) =
            f32x4{acc[cb][a][4 * j], acc[cb][a][4 * j + 1], acc[cb][a][4 * j + 2], acc[cb][a][4 * j + 3]};
      }
  __syncthreads();
  VFX_TS(11);  // staged
  {
    // thread (rt, c4) of the patch loads: keep[q] is x at row rt + 4 q, channels 4 c4 ..
    const f32x4 bv = *(const VFX_GLOBAL f32x4*)(p.b2 + 4 * c4);
    const float aslope = p.act_slope;
    const bool even = (tid & 1) == 0;
    unsigned ya_sat = 0;
    // y and ya through buffer descriptors: 32-bit offsets, and a masked row is an offset beyond the bound (its stores are dropped)
    constexpr unsigned kOob = 0xfffffff0u;  // beyond every descriptor (plan_resblock: tensors < 4 GiB - 4096); nothing is added to it
    const unsigned ybytes = (unsigned)((int64_t)p.B * T * C * 4);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rya = __builtin_amdgcn_make_buffer_rsrc(p.ya, 0, p.ya ? (int)(ybytes / 2) : 0, 0x00020000);
#pragma unroll
    for (int q = 0; q < KEEP; ++q) {
      const int m = rt + q * RPP;  // h pixel of the staged row
      const int li = (int)(((unsigned)m * inv_w1) >> 20), lj = m - li * W1;
      const int pos = base_h + li * rowstride + lj;
      const bool ok = (li < TH) & (lj >= 1) & (lj <= W1 - 2) & ((unsigned)pos < (unsigned)T) & (!p.fold | (j0 + lj - 1 < d));
      const f32x4 val = *reinterpret_cast<const f32x4*>(smem + m * LDO + 4 * c4) + bv + keep[q];  // + x: this thread's own rows
      const unsigned o = (unsigned)(img * T + pos) * (unsigned)(C * 4) + 16u * c4;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), ry, (int)(ok ? o : kOob), 0, 0);
      if (p.ya) {
        // last layer of the stack: also the activated fp16 form for the upsampler that follows (2 bytes per element)
        f32x4 u;
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = fmaxf(val[e], val[e] * aslope);
        const unsigned h01 = pack_f16x2(u[0], u[1], ya_sat), h23 = pack_f16x2(u[2], u[3], ya_sat);
        // quad_perm [1,0,3,2]: the even lane of a pair collects the pair's 8 consecutive channels (16 bytes)
        const unsigned g0 = (unsigned)__builtin_amdgcn_mov_dpp((int)h01, 0xB1, 0xf, 0xf, false);
        const unsigned g1 = (unsigned)__builtin_amdgcn_mov_dpp((int)h23, 0xB1, 0xf, 0xf, false);
        const u32x4 w = {h01, h23, g0, g1};
        __builtin_amdgcn_raw_buffer_store_b128(w, rya, (int)((ok && even) ? o / 2 : kOob), 0, 0);
      }
    }
    if (p.ya) report_f16_saturation(f16_sat_bits_bad(ya_sat), p.flags);
  }
  VFX_TS(12);  // stores issued
  VFX_TS_FLUSH(p.timing, tile, wave_u, NW);
}

int resblock_s256_patch_rows() { return S256_PR; }
int resblock_s256_tile() { return S256_MT; }

// Opt-in (VFX_TUNE_WIDE_SINGLE_FORM): measured against the two-form layer of resblock_w64.hip on the vocoder at 16 x 10 s, this
// layer moves 1.80 GB instead of 2.62 GB per launch and takes 0.76 ms instead of 0.70 (profiles/r03_c9_single_form_ab.jsonl):
// 64-position tiles pay the per-tile phases (parameter loads, address math, 24 loads per thread, h, epilogue: 31 k cycles) twice
// per 128 positions and fetch every weight fragment twice as often.  The default stays the two-form layer.
bool resblock_s256_enabled(int tuning) {
  return (tuning & VFX_TUNE_WIDE_SINGLE_FORM) && !(tuning & (VFX_TUNE_WIDE_8WAVE | VFX_TUNE_NO_FUSED_WIDE | VFX_TUNE_NO_FUSED_STACKS));
}

void launch_resblock_s256(const ResBlockParams& hp, const ResBlockParams* dparams, hipStream_t stream) {
  VFX_CHECK(!hp.asrc && hp.hionly && hp.C == 256 && !hp.geo2d && hp.dil2 == 0 && hp.tile_m == S256_MT && hp.patch_rows == S256_PR,
            "resblock_s256: needs the 16-bit mode, C = 256, %d-position tiles planned with %d patch rows", S256_MT, S256_PR);
  VFX_CHECK((hp.fold ? hp.PW : hp.dil) <= 32, "resblock_s256: the residual window starts beyond patch row 32");
  const int64_t grid = (int64_t)hp.B * hp.tiles_h * hp.tiles_w;
  VFX_CHECK(grid > 0 && grid < ((int64_t)1 << 31), "resblock_s256: bad grid");
  // the staged tile (64 rows x 260 floats) is the largest of the three overlays: 65 KB, two blocks per CU
  constexpr size_t lds = (size_t)S256_MT * (256 + 4) * 4;
  static_assert(lds >= (size_t)(256 / 64) * S256_PR * CROW && lds >= (size_t)S256_MT * 256 * 2 && 2 * lds <= 160 * 1024, "overlays must fit twice");
  static uint64_t attr_devices = 0;
  if (first_use_on_current_device(attr_devices)) {
    VFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_resblock_s256), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  hipLaunchKernelGGL(k_resblock_s256, dim3((int)grid), dim3(256), lds, stream, dparams);
  VFX_HIP(hipGetLastError());
}

}  // namespace vfx
