// resblock_w64.hip -- the fused wide ResStack layer of the 16-bit mode (C = 256)
//
//     y  = x + conv2(LeakyReLU(conv1(xa) + b1)) + b2,      ya = fp16(LeakyReLU_next(y))
//
// as FOUR-wave blocks with TWO blocks per CU.  conv1's operand xa = fp16(LeakyReLU(x)) comes from HBM in MFMA operand form
// (LDS-DMA, no arithmetic).  Two trunk forms:
//   X16 = false (VFX_TUNE_F32_TRUNK, the round-3 layer): x (raw fp32) is read for the residual, y (raw fp32) and ya are written
//       -- 12 bytes per element and layer, which made this layer traffic-bound in practice (2.6 GB per launch);
//   X16 = true (default, round 4): xa is the ONLY form of the trunk.  LeakyReLU with a positive slope is invertible, so the
//       residual is recovered from the operand form itself, x = min(xa, xa / slope) (for xa < 0 the quotient is the smaller one),
//       with the relative precision fp16(x) would have; the epilogue reads the tile's own centre rows of xa a second time (they
//       were fetched for the patch microseconds earlier: L2) and writes ya only -- 4 bytes per element and layer, every byte
//       fetched from HBM is an MFMA operand.
//
// Why four-wave blocks (round-3 measurements): the 8-wave / one-block-per-CU form (k_resblock_act, deleted in round 4) runs a layer in the
// SUM of its memory phases (0.50 ms with all arithmetic removed) and its arithmetic (0.48 ms).  The memory phases are not
// latency chains: a CU streams HBM at ~22 GB/s (10 B/clk) whatever the rest of the chip does, and a tile moves 384 KB -- 17 us
// during which the MFMA pipe idles, because the block that owns the CU's LDS is the one that waits.  And its arithmetic phase is
// operand-bound: every MFMA takes a 1 KB pixel fragment from LDS (wave tile 32 couts x 128 positions).
// Here a wave owns 64 couts x 128 positions (128 accumulator registers): a pixel fragment read from LDS feeds TWO MFMAs, a weight
// fragment four; the block is 4 waves (one per SIMD, <= 256 registers), its LDS fits twice into a CU (80 KB: the xa patch as four
// chunk buffers of 160 rows; h and the staged accumulators overlay them), so two blocks share a CU and the memory phases of one
// run under the arithmetic of the other -- each block has its own waves, hence its own vmcnt queues.  With one wave per SIMD
// nobody hides a wave's own LDS latency, so the compute phases carry no scheduling barriers: the fragment reads of the next K step
// are free to move above the MFMAs of this one (the weight registers are ordered by their `use` statements alone).
//
// Tile geometry: plan_resblock with patch_rows = 160 (1-D tiles for d <= 16, folded rows of d samples above; d = 27 folds here).
#include <type_traits>

#include "conv_common.h"
#include "vfx_internal.h"

namespace vfx {

namespace {
constexpr int W64_PR = 160;  // patch rows per chunk buffer
}

template <int C, bool X16>
__global__ __launch_bounds__(256, 2) void k_resblock_w64(const ResBlockParams* __restrict__ pp) {
  constexpr int NW = 4, NTHR = NW * 64, MT = 128;
  constexpr int NCH = C / 64;                // 64-channel chunks: 128-byte rows of fp16
  static_assert(NCH == NW, "one wave per 64 output channels");
  constexpr int PR = W64_PR;
  constexpr int PBYTES = PR * CROW;
  constexpr int RG = NTHR / 8;               // patch rows per DMA instruction group (8 lanes per row)
  constexpr int NG = PR / RG;                // DMA instructions per wave and chunk
  static_assert(PR % RG == 0 && (RG / 2) % 8 == 0, "patch rows must split into whole DMA groups with one swizzle key");
  constexpr int WM = MT / 32;                // 32-position blocks per wave
  constexpr int WL = 8;                      // weight loads per tap and wave: 2 cout blocks x 4 K steps
  constexpr int HROW = C * 2;                // bytes per h row (fp16)
  constexpr int NT1 = 3 * NCH;               // taps of conv1 (chunk-major); conv2 has as many
  constexpr int EPC = 128, NEP = C / EPC;    // epilogue passes of 128 channels (two waves each)
  constexpr int LDO = EPC + 4;               // staged output row (floats)
  constexpr int V = EPC / 4, RPP = NTHR / V; // 32 float4 per staged row, 8 rows per step
  constexpr int SUB = 4;                     // steps per sub-pass (32 rows): the residual of a sub-pass is requested one ahead
  constexpr int NSUB = MT / (RPP * SUB);     // 4 sub-passes per pass

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
  // tile -> (image, tile row, tile column) with the host's reciprocals (plan_resblock): three integer divisions otherwise
  const int img = div_recip(tile, p.inv_tiles_per_img);
  const int trem = tile - img * (p.tiles_w * p.tiles_h);
  const int ti = div_recip(trem, p.inv_tiles_w);
  const int tj = trem - ti * p.tiles_w;
  const int T = p.T, d = p.dil, W1 = p.W1, TH = p.TH, PW = p.PW, P = p.P;
  const int rowstride = p.fold ? d : 0;
  const int j0 = tj * p.TWo;
  const int base_h = p.fold ? ti * TH * d + j0 - 1 : j0 - 1;  // position of h pixel (0, 0)
  const int base_x = base_h - d;                                // position of patch pixel (0, 0)
  // batches of clips of unequal length (ResBlockParams::lens): this clip's sequence ends at Tb <= T -- positions past it read as
  // zeros, h is zero there, nothing is stored there; a tile wholly past the end has nothing to do
  const int Tb = p.lens ? min(T, ((const VFX_GLOBAL int*)p.lens)[__builtin_amdgcn_readfirstlane(img)] * p.lens_mul) : T;
  if (base_h + 1 >= Tb) return;
  const float slope = p.slope;
  // m / W1 and prow / PW as multiply-shift (rows < 512, divisors <= 320: exact; cf. resblock_rw.hip): an integer division is ~25 VALU
  // instructions, a tile has 25 of them per thread
  const unsigned inv_pw = p.inv_pw, inv_w1 = p.inv_w1;

  const int lr = tid >> 3, cg = tid & 7;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int l31 = lane & 31, lh = lane >> 5;
  const int key_l = (lr >> 1) & 7;  // swizzle key of patch rows lr + RG * q

  // ---- the xa patch: every chunk at once (ONE round trip per tile; the other block of the CU computes meanwhile) -------------
  {
    unsigned voff[NG];
    unsigned okmask = 0;
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      const int prow = lr + RG * q;
      const int pi = (int)(((unsigned)prow * inv_pw) >> 20), pj = prow - pi * PW;
      const int pos = base_x + pi * rowstride + pj;
      const bool ok = (prow < P) & ((unsigned)pos < (unsigned)Tb);
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
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, (int)o, 0, 0, 0);
      }
    }
  }

  VFX_TS(1);  // patch requested
  int arow1[WM];   // A row of this lane's h pixel in the patch (tap offset to be added)
  bool hval[WM];   // that h pixel lies inside the tile's h grid and inside the sequence
#pragma unroll
  for (int a = 0; a < WM; ++a) {
    const int ml = a * 32 + l31;
    const int li = (int)(((unsigned)ml * inv_w1) >> 20), lj = ml - li * W1;
    arow1[a] = li < TH ? li * PW + lj : 0;
    const int pos = base_h + li * rowstride + lj;
    hval[a] = (li < TH) & ((unsigned)pos < (unsigned)Tb);
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

  // ---- pixel fragments: software-pipelined one K step (8 MFMAs = 256 cycles) ahead -----------------------------------------------
  // Step (g, s) = K step s of tap g reads the four fragments of the NEXT step into the other register set before it issues its
  // own eight MFMAs.  rb / kx: byte offset of the lane's row of position block a in the LDS image of tap g, and its swizzle key
  // xor the lane's half (recomputed per tap behind an opaque statement: kept for all taps they would be 200 registers).
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
        for (int i = 0; i < WM; ++i) {  // 1 fragment read, then 2 MFMAs, four times
          if (!last) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
      }
    }
  };

  fetch(0);
  asm volatile("s_waitcnt vmcnt(0)" : : : "memory");  // the patch and tap 0 have landed (this wave's share)
  VFX_TS(2);
  VFX_TS(3);
  __syncthreads();                                     // ... and everybody else's
  VFX_TS(4);

  // ---- phase 1: conv1 (its last tap fetches the first tap of conv2) -------------------------------------------------------------
  conv(prep1, 0, NT1, true);
  // X16 (round 5): the residual never comes from memory a second time.  The tile's centre rows of xa are still in the patch
  // buffers: this wave takes ITS 64 couts x 128 positions (chunk `wave` of the centre tap's rows) in accumulator layout -- lane =
  // pixel, 4 consecutive channels per 8-byte read, 64 registers of packed fp16 -- before the barrier that hands the buffers over
  // to h; phase 2 turns them into x = min(xa, xa / slope) as the INITIAL value of conv2's accumulators.  (Rounds 3-4 re-read the
  // centre rows from L2 in the epilogue: 0.95-1.03 GB read per launch for a 0.40 GB tensor, 12 loads in flight beside the staged
  // tile, profiles/r04_pmc.txt.)
  u32x2 rres[2][WM][4];
  if constexpr (X16) {
#pragma unroll
    for (int a = 0; a < WM; ++a) {
      int r = arow1[a];
      asm volatile("" : "+v"(r));
      const int row = r + poff1;
      const char* rowp = lds + wave_u * PBYTES + row * CROW + 8 * lh;
      const int key = swz_key(row);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int j = 0; j < 4; ++j) rres[cb][a][j] = *reinterpret_cast<const u32x2*>(rowp + (((cb * 4 + j) << 4) ^ key));
    }
  }
  VFX_TS(5);
  __syncthreads();  // every wave is done reading the patch buffers that h overlays
  VFX_TS(6);

  // ---- phase 2: h = LeakyReLU(conv1 + b1) as fp16, zero outside the sequence ---------------------------------------------------
  // Lane (l31, lh) of position block a holds h pixel m = a*32 + l31 and, in registers 4j .. 4j+3 of cout block cb, channels
  // (2w + cb)*32 + 8j + 4lh .. +3: chunk w of the pixel's row, piece cb*4 + j, half lh.
  {
    unsigned sat16 = 0;
    const f16x2 slope2 = {(_Float16)slope, (_Float16)slope};
    const float inv_slope = 1.f / slope;    // X16: the residual x = min(xa, xa / slope)
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
          // convert first, activate the packed halves (conv_common.h: pack_f16x2_sat16 / lrelu_f16x2)
          const unsigned h01 = lrelu_f16x2(pack_f16x2_sat16(acc[cb][a][4 * j] + b1v[j][0], acc[cb][a][4 * j + 1] + b1v[j][1], hval[a], sat16), slope2);
          const unsigned h23 = lrelu_f16x2(pack_f16x2_sat16(acc[cb][a][4 * j + 2] + b1v[j][2], acc[cb][a][4 * j + 3] + b1v[j][3], hval[a], sat16), slope2);
          if constexpr (X16) {  // conv2 accumulates on top of the residual x = min(xa, xa / slope)
            const f32x4 v = f16x4_widen(rres[cb][a][j]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[cb][a][4 * j + e] = fminf(v[e], v[e] * inv_slope);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[cb][a][4 * j + e] = 0.f;
          }
          *reinterpret_cast<uint2*>(rowp + (((cb * 4 + j) ^ key) << 4)) = make_uint2(h01, h23);  // LeakyReLU(0) = 0: masked stays 0
        }
      }
    }
    report_f16_saturation(f16_sat16_bad(sat16), p.flags);
  }
  VFX_TS(7);
  __syncthreads();  // h is complete
  VFX_TS(8);

  // ---- phase 3: conv2 from the resident h (tap NT1 was fetched by conv1's last tap: its wait is the first one of conv()) ----------
  conv(prep2, NT1, 2 * NT1, false);
  VFX_TS(9);

  // ---- phase 4 ------------------------------------------------------------------------------------------------------------------
  if constexpr (X16) {
    // fp16 trunk (round 5): ya = fp16(LeakyReLU(acc + b2)) straight from the accumulators -- the residual is inside them since
    // phase 2 -- with NO staging through LDS and no barrier.  A lane holds, per position block and cout block, four runs of 4
    // consecutive channels (8 bytes of fp16) 16 bytes apart; its partner 32 lanes away holds the runs in between.  One
    // v_permlane32_swap per register pair trades run j + 1 of the lower lane for run j of the upper one: every lane ends up with
    // 16 contiguous bytes and a store instruction writes 32 contiguous bytes of 32 rows.  (Rounds 3-4 staged the tile in LDS in two
    // passes: 256 KB of LDS traffic and four block barriers per tile, 17 k of a block's 80 k cycles, profiles/r04_phase_timing_f16_trunk.txt.)
    int l31e = l31, lhe = lh, we = wave_u;
    asm volatile("" : "+v"(l31e), "+v"(lhe), "+s"(we));  // the epilogue's index math stays behind conv2 (hoisted into its last taps it spills)
    const float aslope = p.act_slope;
    const unsigned yabytes = (unsigned)((int64_t)p.B * T * C * 2);
    const __amdgpu_buffer_rsrc_t rya = __builtin_amdgcn_make_buffer_rsrc(p.ya, 0, (int)yabytes, 0x00020000);
    constexpr unsigned kOob = 0xC0000000u;  // beyond the descriptor (the launch checks the tensor is < 2 GiB), + 1 KB does not wrap
    unsigned ya_sat = 0;
    VFX_TS(10);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      f32x4 b2v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b2v[j] = *(const VFX_GLOBAL f32x4*)(p.b2 + (2 * we + cb) * 32 + 8 * j + 4 * lhe);
#pragma unroll
      for (int a = 0; a < WM; ++a) {
        const int m = a * 32 + l31e;
        const int li = (int)(((unsigned)m * inv_w1) >> 20), lj = m - li * W1;
        const int pos = base_h + li * rowstride + lj;
        const bool ok = (li < TH) & (lj >= 1) & (lj <= W1 - 2) & ((unsigned)pos < (unsigned)Tb) & (!p.fold | (j0 + lj - 1 < d));
        const unsigned rowoff = ok ? (unsigned)(img * T + pos) * (unsigned)(C * 2) + (unsigned)((2 * we + cb) * 64 + 16 * lhe) : kOob;
#pragma unroll
        for (int jp = 0; jp < 4; jp += 2) {
          unsigned q[2][2];  // [run jp, jp + 1][channels 0-1, 2-3]
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            f32x4 u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float v = acc[cb][a][4 * (jp + r) + e] + b2v[jp + r][e];
              u[e] = fmaxf(v, v * aslope);
            }
            q[r][0] = pack_f16x2(u[0], u[1], ya_sat);
            q[r][1] = pack_f16x2(u[2], u[3], ya_sat);
          }
          // lanes 0-31 keep their run jp and receive the partner's run jp; lanes 32-63 receive the partner's run jp + 1 and keep theirs
          const auto s0 = __builtin_amdgcn_permlane32_swap(q[0][0], q[1][0], false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(q[0][1], q[1][1], false, false);
          const u32x4 w = {s0[0], s1[0], s0[1], s1[1]};
#ifdef VFX_ABL_NOSTORE  // timing-only build (wrong results): what the direct stores cost
          asm volatile("" : : "v"(w));
#else
          __builtin_amdgcn_raw_buffer_store_b128(w, rya, (int)(rowoff + (unsigned)(16 * jp)), 0, 0);
#endif
        }
      }
    }
    VFX_TS(11);
    report_f16_saturation(f16_sat_bits_bad(ya_sat), p.flags);
  } else {
  // ---- phase 4: y = conv2 + b2 + x, raw fp32 and (optionally) activated fp16 -----------------------------------------------------
  // Two passes of 128 channels (the pass of waves 0, 1, then the one of waves 2, 3) through a staged tile in LDS; every pass in
  // NSUB sub-passes of 32 rows whose residual is requested one sub-pass ahead (the loads of sub-pass s + 1 are issued BEFORE
  // the stores of sub-pass s, so no load waits behind a store: vmcnt counts both, in order).
  int tid_e = tid;
  asm volatile("" : "+v"(tid_e));  // the epilogue's index math stays behind conv2 (hoisted into its last taps it spills)
  const int c4 = tid_e % V, r0 = tid_e / V;
  const bool even = (tid_e & 1) == 0;
  const float aslope = p.act_slope;
  // Residual, y and ya go through buffer descriptors: 32-bit byte offsets (half the address registers of 64-bit pointers: the
  // epilogue runs beside the 128 accumulators of the waves that stage second), and a masked row is simply an offset beyond the
  // descriptor's bound -- its load returns zeros, its stores are dropped.
  const unsigned ybytes = (unsigned)((int64_t)p.B * T * C * 4);
  // X16: the residual comes from xa itself (fp16); there is no x and no y
  const __amdgpu_buffer_rsrc_t rx = X16 ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.xa), 0, (int)(ybytes / 2), 0x00020000)
                                        : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, X16 ? 0 : (int)ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rya = __builtin_amdgcn_make_buffer_rsrc(p.ya, 0, p.ya ? (int)(ybytes / 2) : 0, 0x00020000);
  constexpr unsigned kOob = 0xC0000000u;  // beyond every descriptor (the launch checks the tensors are < 2 GiB), + 1 KB does not wrap
  constexpr unsigned EB = X16 ? 2u : 4u;  // bytes per element of the tensors `ooff` addresses
  unsigned ooff[NSUB][SUB];  // byte offset of the row's first channel in y (fp32; ya: half of it) -- X16: in xa / ya (fp16)
#pragma unroll
  for (int sp = 0; sp < NSUB; ++sp)
#pragma unroll
    for (int q = 0; q < SUB; ++q) {
      const int m = r0 + (sp * SUB + q) * RPP;  // h pixel of the staged row
      const int li = (int)(((unsigned)m * inv_w1) >> 20), lj = m - li * W1;
      const int pos = base_h + li * rowstride + lj;
      const bool ok = (li < TH) & (lj >= 1) & (lj <= W1 - 2) & ((unsigned)pos < (unsigned)Tb) & (!p.fold | (j0 + lj - 1 < d));
      ooff[sp][q] = ok ? (unsigned)(img * T + pos) * (unsigned)(C * EB) : kOob;
    }
  // The residual runs THREE sub-passes (12 loads, 48 registers) ahead: the first three are requested here, before the staging
  // barriers, sub-pass i + 3 as soon as sub-pass i has consumed its registers.  (Round 3, phase stamps: with the residual
  // requested ONE sub-pass ahead -- and the compiler moving that request behind the stores -- the epilogue was eight exposed
  // memory latencies: 32 k of a block's 91 k cycles.  A whole pass ahead (64 registers) spills beside the 128 accumulators.)
#ifndef VFX_W64_NRB
#define VFX_W64_NRB 3
#endif
  // residual register sets: sub-pass i uses set i % NRB and then requests sub-pass i + NRB into it (X16: a set is 8 registers
  // instead of 16; -DVFX_W64_NRB=4 = a whole pass ahead, measured in round 4: profiles/r04_*)
  constexpr int NRB = X16 ? VFX_W64_NRB : 3;
  typedef typename std::conditional<X16, u32x2, u32x4>::type res_t;  // 4 channels: fp16 (8 bytes) or fp32 (16 bytes)
  res_t res[NRB][SUB];
  auto request_res = [&](int idx) __attribute__((always_inline)) {  // idx = pass * NSUB + sub-pass
    const int pass = idx / NSUB, sp = idx % NSUB;
#pragma unroll
    for (int q = 0; q < SUB; ++q) {
      const int o = (int)(ooff[sp][q] + (unsigned)(pass * EPC + 4 * c4) * EB);
      if constexpr (X16) res[idx % NRB][q] = __builtin_amdgcn_raw_buffer_load_b64(rx, o, 0, 0);
      else res[idx % NRB][q] = __builtin_amdgcn_raw_buffer_load_b128(rx, o, 0, 0);
    }
  };
  // the residual of row q of the register set: X16: x = min(xa, xa / slope) of the fp16 operand form (a masked row loads zeros)
  // (X16: the residual is already inside the accumulators -- see phase 2 -- and nothing is requested here)
  auto res_f32 = [&](const res_t& r) __attribute__((always_inline)) -> f32x4 {
    if constexpr (X16) return f32x4{0.f, 0.f, 0.f, 0.f};
    else return __builtin_bit_cast(f32x4, r);
  };
  if constexpr (!X16) {
#pragma unroll
    for (int i = 0; i < NRB; ++i) request_res(i);
  }
  __syncthreads();  // every wave is done with h
  VFX_TS(10);
  unsigned ya_sat = 0;
#pragma unroll
  for (int pass = 0; pass < NEP; ++pass) {
    if ((wave_u >> 1) == pass) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int a = 0; a < WM; ++a)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int row = a * 32 + l31;
            *reinterpret_cast<f32x4*>(smem + row * LDO + (wave_u & 1) * 64 + cb * 32 + 8 * j + 4 * lh) =
                f32x4{acc[cb][a][4 * j], acc[cb][a][4 * j + 1], acc[cb][a][4 * j + 2], acc[cb][a][4 * j + 3]};
          }
    }
    __syncthreads();  // the pass is staged
    const int ncol = pass * EPC + 4 * c4;
    const f32x4 bv = *(const VFX_GLOBAL f32x4*)(p.b2 + ncol);
#pragma unroll
    for (int sp = 0; sp < NSUB; ++sp) {
      f32x4 val[SUB];
#pragma unroll
      for (int q = 0; q < SUB; ++q)
        val[q] = *reinterpret_cast<const f32x4*>(smem + (r0 + (sp * SUB + q) * RPP) * LDO + 4 * c4) + bv +
                 res_f32(res[(pass * NSUB + sp) % NRB][q]);
      if constexpr (!X16)
        if (pass * NSUB + sp + NRB < NEP * NSUB) request_res(pass * NSUB + sp + NRB);  // into the registers just consumed
      if constexpr (!X16) {
#pragma unroll
        for (int q = 0; q < SUB; ++q)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val[q]), ry, (int)(ooff[sp][q] + (unsigned)(ncol * 4)), 0, 0);
      }
      if (X16 || p.ya) {
#pragma unroll
        for (int q = 0; q < SUB; ++q) {
          f32x4 u;
#pragma unroll
          for (int e = 0; e < 4; ++e) u[e] = fmaxf(val[q][e], val[q][e] * aslope);
          const unsigned h01 = pack_f16x2(u[0], u[1], ya_sat), h23 = pack_f16x2(u[2], u[3], ya_sat);
          // quad_perm [1,0,3,2]: the even lane of a pair collects the pair's 8 consecutive channels (16 bytes)
          const unsigned g0 = (unsigned)__builtin_amdgcn_mov_dpp((int)h01, 0xB1, 0xf, 0xf, false);
          const unsigned g1 = (unsigned)__builtin_amdgcn_mov_dpp((int)h23, 0xB1, 0xf, 0xf, false);
          const u32x4 w = {h01, h23, g0, g1};
          // (half of a masked row's offset could land INSIDE ya: the odd lanes and the masked rows get the out-of-bounds offset itself)
          const unsigned ao = (even && ooff[sp][q] != kOob) ? ooff[sp][q] / (EB / 2) + (unsigned)(ncol * 2) : kOob;
          __builtin_amdgcn_raw_buffer_store_b128(w, rya, (int)ao, 0, 0);
        }
      }
    }
    if (pass == 0) { VFX_TS(11); }  // pass 0 stored
    if (pass + 1 < NEP) __syncthreads();  // the staged pass has been consumed
  }
  if (p.ya) report_f16_saturation(f16_sat_bits_bad(ya_sat), p.flags);
  }
  VFX_TS(12);
  VFX_TS_FLUSH(p.timing, tile, wave_u, NW);
}

int resblock_w64_patch_rows() { return W64_PR; }

bool resblock_w64_supported(int C) { return C == 256; }

void launch_resblock_w64(const ResBlockParams& hp, const ResBlockParams* dparams, hipStream_t stream) {
  VFX_CHECK(hp.asrc && hp.hionly && hp.C == 256 && hp.xa && hp.tile_m == 128 && hp.patch_rows == W64_PR,
            "resblock_w64: needs the 16-bit mode, C = 256, 128-position tiles planned with %d patch rows", W64_PR);
  VFX_CHECK(hp.x16 ? (hp.ya && !hp.x && !hp.y && hp.slope > 0.f) : (hp.x && hp.y),
            "resblock_w64: %s", hp.x16 ? "the fp16 trunk is the activated tensor alone (ya, no x / y) and needs an invertible LeakyReLU (slope > 0)"
                                       : "the two-form trunk needs x and y");
  const int64_t grid = (int64_t)hp.B * hp.tiles_h * hp.tiles_w;
  VFX_CHECK(grid > 0 && grid < ((int64_t)1 << 31), "resblock_w64: bad grid");
  // the epilogue addresses x / y / ya with 32-bit offsets and masks rows with an offset of 3 GiB
  VFX_CHECK((int64_t)hp.B * hp.T * hp.C * 4 < ((int64_t)1 << 31), "resblock_w64: tensor exceeds 2 GiB");
  // 4 chunk buffers of 160 rows = 80 KB (h: 64 KB and the staged pass: 66 KB overlay them): exactly two blocks per CU
  const size_t lds = (size_t)(256 / 64) * W64_PR * CROW;
  static_assert((256 / 64) * W64_PR * CROW >= 128 * (128 + 4) * 4 && (256 / 64) * W64_PR * CROW >= 128 * 256 * 2, "overlays must fit");
  static uint64_t attr_devices = 0;
  if (first_use_on_current_device(attr_devices)) {
    VFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_resblock_w64<256, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    VFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_resblock_w64<256, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  if (hp.x16) hipLaunchKernelGGL((k_resblock_w64<256, true>), dim3((int)grid), dim3(256), lds, stream, dparams);
  else hipLaunchKernelGGL((k_resblock_w64<256, false>), dim3((int)grid), dim3(256), lds, stream, dparams);
  VFX_HIP(hipGetLastError());
}

}  // namespace vfx
