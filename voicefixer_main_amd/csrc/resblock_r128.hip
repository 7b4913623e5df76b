// resblock_r128.hip -- one fused TFGAN ResStack layer of the 16-bit mode at C = 128 (raw trunk: fp16, or fp32 with
// VFX_TUNE_F32_TRUNK; cf. resblock.hip)
//
//     y = x + conv2(LeakyReLU(conv1(LeakyReLU(x)) + b1)) + b2        conv1: k3, dilation d;  conv2: k3, dilation 1
//
// as FOUR-wave blocks, two per CU, that read x ONCE.
//
// Why (round-3 PMC, profiles/r03_*): k_resblock<128, 8> moves 3.6-3.7 GB per layer through the fabric for 2.42 GB of tensors
// -- the epilogue's re-read of the residual finds its lines evicted (2.35-2.7 GB read for a 1.21 GB tensor) -- at ~4.1 TB/s:
// the stack is bandwidth-bound on bytes it does not need.  Here every thread keeps the raw values of the tile's 128 centre rows
// that it fetched for the patch (64 registers; a block is 4 waves = one per SIMD with 256 registers each) and adds them to the
// staged conv2 result in LDS: the epilogue issues no load at all.  The patch goes global -> registers -> fp16 operand rows in LDS
// (4 chunk buffers of 160 rows x 128 B = 80 KB = half a CU's LDS; h and the staged tile overlay them), so two blocks share a CU
// and the memory phases of one run under the arithmetic of the other; a wave owns 64 couts x
// 64 positions: a pixel fragment (1 KB from LDS) feeds TWO MFMAs and a weight fragment two -- with 32 couts x 128 positions per
// wave every MFMA took its own pixel fragment, 128 B/clk/CU at full MFMA rate = all the LDS can deliver, and the convolution
// phases ran 2.3x their MFMA time (phase stamps, profiles/r03_phase_timing_*.txt).  The weight ring runs three taps ahead (a
// tap is only 8 MFMAs here); the pixel fragments are software-pipelined one K step ahead; no scheduling barriers and no memory
// clobbers in the compute phases.
//
// Tile geometry: plan_resblock with patch_rows = 160 (1-D tiles for d <= 16, folded rows of d samples with 14-wide tiles above).
// Weights: pack_conv mode 2 (fp16 in the hi fragments of 32-channel chunks).
//
// X16 (round 4, default): the trunk between two launches is an fp16 tensor (2 bytes per element).  The layer was byte-bound on
// an fp32 trunk whose values were rounded to fp16 before every MFMA anyway (2.43 GB per layer at 3.1 TB/s); now a thread's load
// of 4 channels is 8 bytes, the operand is formed from the packed halves (v_pk_mul_f16 + v_pk_max_f16: no conversion, nothing to
// saturate), the residual is widened to fp32 once and kept in the same 64 registers, the sum is formed in fp32 and rounded once
// when it is stored (saturation flagged).  A pair's intermediate tensor stays fp32 in registers as before.
#include <type_traits>

#include "conv_common.h"
#include "vfx_internal.h"

namespace vfx {

namespace {
constexpr int R128_PR = 160;  // patch rows per chunk buffer
}

// PAIR: TWO consecutive layers (dilations d, d2: the plan pairs (1, 3)) in one pass -- the first layer's output y1 never leaves
// the CU: raw, it replaces x in the 64 registers that hold the residual; activated, it is written as the operand patch of the
// second layer over the first one's.  Both layers work over the 128-index space of the tile: y1 is valid on indices 1 .. 126, the
// second layer's h on 1 + d2 .. 126 - d2, the outputs on 2 + d2 .. 125 - d2 (plan_resblock: 128 - 4 - 2 d2 positions per tile).
// Measured and NOT kept (round 6, profiles/r06_c15_voc_layers_r128_persistent_loser.txt): the single layers as PERSISTENT blocks (two
// per CU, each walking its XCD's tiles, set-up paid once per block): 3.94 -> 4.14 ms for the six layers on one box, alternating.  A
// fresh workgroup's set-up runs beside the other block's arithmetic either way; the loop only adds its barrier and registers.
template <bool PAIR, bool X16>
__global__ __launch_bounds__(256, 2) void k_resblock_r128(const ResBlockParams* __restrict__ pp) {
  constexpr int C = 128, NW = 4, NTHR = NW * 64, MT = 128;
  constexpr int NCH = C / 32;                // 32-channel chunks = waves along the couts
  constexpr int PR = R128_PR;
  constexpr int PBYTES = PR * CROW;
    static_assert(PR % 8 == 0 && NCH == 4 && NW == 4, "geometry");
  constexpr int WM = 2, WN = 2;              // 32-position blocks / 32-cout blocks per wave: 64 positions x 64 couts
  constexpr int WL = 2 * WN;                 // weight loads per tap and wave (the hi fragments f[0], f[2] of two cout blocks): the vmcnt counts below
  static_assert(WL == 4, "the hand-counted waits of conv() assume four loads per tap");
  // weight taps in flight: a tap is 8 MFMAs (256 cycles), an L2 round trip ~3 of them (a pair keeps one slot less: registers)
  constexpr int RING = PAIR ? 3 : 4, AHEAD = RING - 1;
  constexpr int HROW = C * 4;                // bytes per h row (operand form: 128-byte chunk rows, fp16 in the first half)
  constexpr int NT1 = 3 * NCH;               // taps of conv1 (chunk-major); conv2 has as many
  constexpr int NTOT = (PAIR ? 4 : 2) * NT1; // taps of the launch
  constexpr int LDO = C + 4;                 // staged output row (floats)
  constexpr int KEEP = MT / 8;               // centre rows per thread (16: rows rt + 8 j of one chunk)
  // DIRECT (round 5, the fp16 trunk): no staged tile and no residual registers in store layout.  The patch rows use only the
  // logical pieces 0..3 of their 128-byte rows (32 channels of fp16); the RAW fp16 values of the centre rows are parked in pieces
  // 4..7 of the same rows.  After conv1 every wave reads its 64 couts x 64 positions of them back in ACCUMULATOR layout (before the
  // barrier that hands the buffers over to h) and conv2 accumulates on top of the residual; the epilogue adds b2 and stores straight
  // from the accumulators (v_permlane32_swap pairs a lane's 8-byte runs with its partner's into 16-byte stores, cf.
  // resblock_w64.hip).  A pair keeps y1 = the first layer's output in 64 registers in accumulator layout: activated, it is
  // written as the second layer's operand rows; raw, it is the initial value of the second conv2's accumulators.
  constexpr bool DIRECT = X16;

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
  const int d2 = PAIR ? p.dil2 : 0;
  // position of h pixel (0, 0); a pair's tile starts 1 + d2 further left (the second layer's halo)
  const int base_h = PAIR ? j0 - 2 - d2 : (p.fold ? ti * TH * d + j0 - 1 : j0 - 1);
  const int base_x = base_h - d;                                // position of patch pixel (0, 0)
  // batches of clips of unequal length (ResBlockParams::lens): this clip's sequence ends at Tb <= T -- positions past it read as
  // zeros, h (and a pair's intermediate tensor) is zero there, nothing is stored there; a tile wholly past the end has nothing to do
  const int Tb = p.lens ? min(T, ((const VFX_GLOBAL int*)p.lens)[__builtin_amdgcn_readfirstlane(img)] * p.lens_mul) : T;
  if ((PAIR ? j0 : base_h + 1) >= Tb) return;
  const float slope = p.slope;
  // m / W1 and prow / PW as multiply-shift (rows < 512, divisors <= 320: exact; cf. resblock_rw.hip): an integer division is ~25 VALU
  // instructions, a tile has 25 of them per thread
  const unsigned inv_pw = p.inv_pw, inv_w1 = p.inv_w1;

  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int l31 = lane & 31, lh = lane >> 5;
  const int wn = wave_u & 1, wm = wave_u >> 1;  // cout half / position half of this wave

  int arow1[WM];   // A row of this lane's h pixel in the patch (tap offset to be added)
  bool hval[WM];   // that h pixel lies inside the tile's h grid and inside the sequence
#pragma unroll
  for (int a = 0; a < WM; ++a) {
    const int ml = wm * 64 + a * 32 + l31;
    const int li = (int)(((unsigned)ml * inv_w1) >> 20), lj = ml - li * W1;
    arow1[a] = li < TH ? li * PW + lj : 0;
    const int pos = base_h + li * rowstride + lj;
    hval[a] = (li < TH) & ((unsigned)pos < (unsigned)Tb);
  }
  // weights: (32-channel chunk, tap) blocks of C / 32 cout blocks x 1024 floats; this wave's cout blocks are 2 wn, 2 wn + 1
  const unsigned nb_off = (unsigned)(2 * wn * 1024 + lane * 4) * 4u;
  const unsigned nb_off2 = nb_off + 4096u;
  const int64_t ts = (int64_t)C * kKC;
  const float* const w1p = p.w1;
  const float* const w2p = p.w2;
  const float* const w1bp = PAIR ? p.w1b : nullptr;
  const float* const w2bp = PAIR ? p.w2b : nullptr;
  const int poff0 = p.poff[0], poff1 = p.poff[1], poff2 = p.poff[2];

  f32x16 acc[WN][WM];
#pragma unroll
  for (int n = 0; n < WN; ++n)
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][a][r] = 0.f;

  // ---- weight ring: global tap g in register pair g % RING, AHEAD taps ahead (no "memory" clobbers in the compute phases) -------
  // ring slot s: (k 0..15, k 16..31) of cout block 2 wn in Wa, Wb and of cout block 2 wn + 1 in Wc, Wd
  f32x4 Wa0 = {}, Wb0 = {}, Wc0 = {}, Wd0 = {}, Wa1 = {}, Wb1 = {}, Wc1 = {}, Wd1 = {};
  f32x4 Wa2 = {}, Wb2 = {}, Wc2 = {}, Wd2 = {}, Wa3 = {}, Wb3 = {}, Wc3 = {}, Wd3 = {};
  auto load_w = [&](f32x4& A, f32x4& B, f32x4& Cc, f32x4& D, const float* wtap) __attribute__((always_inline)) {
    asm volatile(
        "s_nop 4\n\t"
        "global_load_dwordx4 %0, %4, %6\n\t"
        "global_load_dwordx4 %1, %4, %6 offset:2048\n\t"
        "global_load_dwordx4 %2, %5, %6\n\t"
        "global_load_dwordx4 %3, %5, %6 offset:2048"
        : "=&v"(A), "=&v"(B), "=&v"(Cc), "=&v"(D)
        : "v"(nb_off), "v"(nb_off2), "s"(wtap));
  };
  auto fetch = [&](int g) __attribute__((always_inline)) {
    const float* w = g < NT1 ? w1p + g * ts : (g < 2 * NT1 ? w2p + (g - NT1) * ts : (g < 3 * NT1 ? w1bp + (g - 2 * NT1) * ts : w2bp + (g - 3 * NT1) * ts));
    switch (g % RING) {
      case 0: load_w(Wa0, Wb0, Wc0, Wd0, w); break;
      case 1: load_w(Wa1, Wb1, Wc1, Wd1, w); break;
      case 2: load_w(Wa2, Wb2, Wc2, Wd2, w); break;
      default: load_w(Wa3, Wb3, Wc3, Wd3, w); break;
    }
  };
  // The registers of a ring slot are readable behind this statement (a counted s_waitcnt precedes it in program order: asm
  // volatile statements keep their order); every reader depends on its outputs.
  auto use_slot = [&](int s) __attribute__((always_inline)) {
    switch (s) {
      case 0: asm volatile("" : "+v"(Wa0), "+v"(Wb0), "+v"(Wc0), "+v"(Wd0)); break;
      case 1: asm volatile("" : "+v"(Wa1), "+v"(Wb1), "+v"(Wc1), "+v"(Wd1)); break;
      case 2: asm volatile("" : "+v"(Wa2), "+v"(Wb2), "+v"(Wc2), "+v"(Wd2)); break;
      default: asm volatile("" : "+v"(Wa3), "+v"(Wb3), "+v"(Wc3), "+v"(Wd3)); break;
    }
  };

#pragma unroll
  for (int g = 0; g < AHEAD; ++g) fetch(g);  // issued BEFORE the patch loads: they stay in flight through the transform

  // ---- the x patch: global -> registers -> fp16 operand rows in LDS; the 128 centre rows stay in registers as the residual --------
  // Patch row pr of chunk c: 32 floats.  Thread (rt, ct, cgt) = (tid / 32, chunk, 4-float piece) -- the thread that will STORE
  // channels 32 ct + 4 cgt .. + 3 of the output rows m = rt + 8 j in the epilogue -- loads, of chunk ct, the x samples of exactly
  // those rows (patch rows m + off, off = d for 1-D tiles, PW = one patch row up for folded ones: the centre window of the
  // patch) and keeps their raw values: the residual is added in the output pass from registers, no thread ever hands it to
  // another one, and x is read from memory ONCE.  The up to 32 halo rows around the window are shared out four per thread.  A
  // wave's load instruction covers two whole 512-byte rows of x.  (Round 3 first staged the raw patch in LDS by LDS-DMA and
  // transformed it in place: 20 DMA instructions per wave at 100-185 issue cycles each, a barrier, and a read + write pass over
  // 80 KB of LDS -- 15 k of a block's 40 k cycles before the first MFMA, phase stamps in profiles/r03_phase_timing_*.txt.)
  // Operand form (k_conv's 16-bit rows): the 8 bytes of the four channels at slot ((cgt >> 1) ^ key(row)), half cgt & 1, of the
  // chunk's 128-byte row.  Rows outside the sequence / the patch are loaded beyond the descriptor's bound: zeros, and
  // LeakyReLU(0) = 0.
  const int off = p.fold ? PW : d;  // <= 32 (launch_resblock_r128)
  const int rt = tid >> 5, ct = (tid >> 3) & 3, cgt = tid & 7;
  constexpr int NROW = PR / 8;      // rows per thread: 16 centre + 4 halo
  f32x4 keep[KEEP];
  if constexpr (X16) {
    // fp16 trunk (round 6): a thread's row piece is 16 bytes = 8 channels (16 lanes per 256-byte row), ten loads per thread instead of
    // twenty, through a descriptor that covers exactly the tile's CLIP -- positions in front of / behind the clip are out of range
    // and arrive as zeros, no compare per row -- and one ds_write_b128 per operand piece / raw piece (the phase stamps and
    // instruction counts of round 6 say the block is bound by the instructions it issues: the request + transform of the 8-byte
    // form were 530 VALU instructions per wave and tile).
    const int rt16 = tid >> 4, c16 = tid & 15;
    const int ct2 = c16 >> 2, pc = c16 & 3;  // 32-channel chunk, 16-byte piece inside the chunk's 64 bytes of operands
    constexpr int NROW2 = PR / 16;           // 8 centre rows + 2 halo rows per thread
    constexpr unsigned kOobL = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.x) + (int64_t)img * T * (C * 2)), 0, Tb * (C * 2), 0x00020000);
    int prow[NROW2];
    u32x4 raw[NROW2];
#pragma unroll
    for (int j = 0; j < NROW2; ++j) {
      const int hh = rt16 + 16 * (j - MT / 16);
      prow[j] = j < MT / 16 ? rt16 + 16 * j + off : (hh < off ? hh : hh + MT);
      const int pi = (int)(((unsigned)prow[j] * inv_pw) >> 20), pj = prow[j] - pi * PW;
      const int pos = base_x + pi * rowstride + pj;
      const unsigned o = prow[j] < P ? (unsigned)pos * (unsigned)(C * 2) + 16u * c16 : kOobL;
      raw[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)o, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);  // all ten loads are in flight before the first row is converted
    VFX_TS(1);  // patch requested
#pragma unroll
    for (int j = 0; j < NROW2; ++j) {
      const int key = (prow[j] >> 1) & 7;
      const int at = ct2 * PBYTES + prow[j] * CROW + ((pc ^ key) << 4);
      // DIRECT: the raw piece beside its operand form, logical piece 4 + pc of the same row (slot (4 + pc) ^ key = the operand's slot ^ 4)
      if (j < MT / 16) *reinterpret_cast<u32x4*>(lds + (at ^ 64)) = raw[j];
      const u32x2 lo = f16x4_lrelu(u32x2{raw[j][0], raw[j][1]}, slope), hi = f16x4_lrelu(u32x2{raw[j][2], raw[j][3]}, slope);
      *reinterpret_cast<u32x4*>(lds + at) = u32x4{lo.x, lo.y, hi.x, hi.y};
    }
  } else {
    unsigned f16_sat = 0;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)(unsigned)((int64_t)p.B * T * C * 4u), 0x00020000);
    const unsigned lane_off = (unsigned)(ct * 32 + cgt * 4) * 4u;
    int prow[NROW];
    u32x4 raw[NROW];
#pragma unroll
    for (int j = 0; j < NROW; ++j) {
      const int hh = rt + 8 * (j - KEEP);
      prow[j] = j < KEEP ? rt + 8 * j + off : (hh < off ? hh : hh + MT);
      const int pi = (int)(((unsigned)prow[j] * inv_pw) >> 20), pj = prow[j] - pi * PW;
      const int pos = base_x + pi * rowstride + pj;
      const bool ok = (prow[j] < P) & ((unsigned)pos < (unsigned)Tb);
      const unsigned o = ok ? (unsigned)(img * T + pos) * (unsigned)(C * 4u) + lane_off : 0xfffffff0u;
      raw[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)o, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);  // all twenty loads are in flight before the first row is converted
    VFX_TS(1);  // patch requested
#pragma unroll
    for (int j = 0; j < NROW; ++j) {
      const int key = (prow[j] >> 1) & 7;
      uint2* const dst = reinterpret_cast<uint2*>(lds + ct * PBYTES + prow[j] * CROW + (((cgt >> 1) ^ key) << 4) + 8 * (cgt & 1));
      const f32x4 r = __builtin_bit_cast(f32x4, raw[j]);
      if (j < KEEP) keep[j] = r;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(r[e], r[e] * slope);
      *dst = make_uint2(pack_f16x2(v[0], v[1], f16_sat), pack_f16x2(v[2], v[3], f16_sat));
    }
    report_f16_saturation(f16_sat_bits_bad(f16_sat), p.flags);
  }
  VFX_TS(2);  // loaded and transformed
  VFX_TS(3);
  __syncthreads();  // the operand rows of every wave are visible
  VFX_TS(4);

  // ---- pixel fragments: software-pipelined one K step (4 MFMAs) ahead ------------------------------------------------------------
  int rb[2][WM], kx[2][WM];
  auto prep1 = [&](int g) __attribute__((always_inline)) {  // conv1: the patch chunk of tap g, rows arow1 + tap offset
    const bool second = g >= 2 * NT1;  // a pair's second layer: y1a row of index m sits at patch row m + d2, tap k reads m + k d2
    const int gl = second ? g - 2 * NT1 : g;
    const int c = gl / 3, k = gl % 3;
#pragma unroll
    for (int a = 0; a < WM; ++a) {
      int r = second ? wm * 64 + a * 32 + l31 : arow1[a];
      if constexpr (PAIR || !X16) asm volatile("" : "+v"(r));  // per-tap addresses are recomputed, not kept (single layers on the fp16 trunk have the registers: the six (tap, block) rows are computed once)
      const int row = r + (second ? k * d2 : (k == 0 ? poff0 : (k == 1 ? poff1 : poff2)));
      rb[g & 1][a] = c * PBYTES + row * CROW;
      kx[g & 1][a] = swz_key(row) ^ (16 * lh);
    }
  };
  auto prep2 = [&](int g) __attribute__((always_inline)) {  // conv2: h rows m + k - 1; chunk c of row r sits at chunk position c ^ (r & 1)
    const int gl = g >= 3 * NT1 ? g - 3 * NT1 : g - NT1;
    const int c = gl / 3, k = gl % 3;
    int lrow = l31;
    if constexpr (PAIR || !X16) asm volatile("" : "+v"(lrow));
#pragma unroll
    for (int a = 0; a < WM; ++a) {
      const int r0 = wm * 64 + a * 32 + lrow + k - 1;
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
  // K step st (0, 1) of tap g: D = W (A operand: rows = couts) x image rows (B operand: columns = pixels)
  auto mm = [&](const f16x8 (&px)[WM], int g, int st) __attribute__((always_inline)) {
    f32x4 w0, w1;
    switch (g % RING) {
      case 0: w0 = st ? Wb0 : Wa0; w1 = st ? Wd0 : Wc0; break;
      case 1: w0 = st ? Wb1 : Wa1; w1 = st ? Wd1 : Wc1; break;
      case 2: w0 = st ? Wb2 : Wa2; w1 = st ? Wd2 : Wc2; break;
      default: w0 = st ? Wb3 : Wa3; w1 = st ? Wd3 : Wc3; break;
    }
    const f16x8 wf0 = __builtin_bit_cast(f16x8, w0), wf1 = __builtin_bit_cast(f16x8, w1);
#pragma unroll
    for (int a = 0; a < WM; ++a) acc[0][a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf0, px[a], acc[0][a], 0, 0, 0);
#pragma unroll
    for (int a = 0; a < WM; ++a) acc[1][a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf1, px[a], acc[1][a], 0, 0, 0);
  };
  // taps g0 .. g1-1 of one convolution; `more`: the launch has taps behind g1 (conv1: conv2's), fetched ahead from here
  auto conv = [&](auto prep, int g0, int g1, bool more) __attribute__((always_inline)) {
    prep(g0);
    rd(pxE, g0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, WM, 0);  // the prologue's reads are a group of their own: the pattern below starts behind them
#pragma unroll
    for (int g = g0; g < g1; ++g) {
      const int gmax = more ? NTOT : g1;  // taps that exist
      // fetch tap g + AHEAD if it exists; then tap g's loads are older than the fetches issued after them
      const int younger = (g + AHEAD < gmax ? AHEAD : gmax - 1 - g);
      if (g + AHEAD < gmax) fetch(g + AHEAD);
      if (younger == 3) asm volatile("s_waitcnt vmcnt(12)");
      else if (younger == 2) asm volatile("s_waitcnt vmcnt(8)");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)");
      else asm volatile("s_waitcnt vmcnt(0)");
      use_slot(g % RING);
      if (g + 1 < g1) prep(g + 1);
      // step 0 (fragments in pxE): read step 1 of this tap; step 1 (pxO): read step 0 of the next tap
      // (the two reads of the next step FIRST, then this step's four MFMAs: a step is only 128 cycles, about one LDS latency --
      // interleaved one by one, the youngest read would be 32 cycles old when the next step waits for it)
      rd(pxO, g, 1);
      mm(pxE, g, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, WM, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, WM * WN, 0);
      const bool last = g + 1 == g1;
      if (!last) rd(pxE, g + 1, 0);
      mm(pxO, g, 1);
      if (!last) __builtin_amdgcn_sched_group_barrier(0x100, WM, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, WM * WN, 0);
    }
  };

  // h = LeakyReLU(conv1 + bias) in operand form, zero where `valid` says the h pixel lies outside the tile's grid / the sequence.
  // Lane (l31, lh) of position block a holds h pixel m = 64 wm + 32 a + l31 and, in registers 4j .. 4j+3 of cout block n, channels
  // (2 wn + n) * 32 + 8j + 4lh .. +3: chunk 2 wn + n of the pixel's row, piece j, half lh.
  // DIRECT: conv2's accumulators start from the residual -- `init` = 1: the raw fp16 pieces `rres` read out of the patch rows, 2: y1
  u32x2 rres[WN][WM][4];
  f32x16 y1[WN][WM];
  auto write_h = [&](const float* bias, const bool (&valid)[WM], auto init_tag) __attribute__((always_inline)) {
    constexpr int INIT = decltype(init_tag)::value;
    unsigned sat16 = 0;
    const f16x2 slope2 = {(_Float16)slope, (_Float16)slope};
#pragma unroll
    for (int n = 0; n < WN; ++n) {
      const int ch = 2 * wn + n;
      f32x4 b1v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b1v[j] = *(const VFX_GLOBAL f32x4*)(bias + ch * 32 + 8 * j + 4 * lh);
#pragma unroll
      for (int a = 0; a < WM; ++a) {
        const int m = wm * 64 + a * 32 + l31;
        char* rowp = lds + m * HROW + (ch ^ (m & 1)) * CROW + 8 * lh;  // chunk parity swap: see prep2()
        const int key = (m >> 1) & 7;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // convert first, activate the packed halves (conv_common.h: pack_f16x2_sat16 / lrelu_f16x2)
          const unsigned h01 = lrelu_f16x2(pack_f16x2_sat16(acc[n][a][4 * j] + b1v[j][0], acc[n][a][4 * j + 1] + b1v[j][1], valid[a], sat16), slope2);
          const unsigned h23 = lrelu_f16x2(pack_f16x2_sat16(acc[n][a][4 * j + 2] + b1v[j][2], acc[n][a][4 * j + 3] + b1v[j][3], valid[a], sat16), slope2);
          if constexpr (INIT == 1) {
            const f32x4 v = f16x4_widen(rres[n][a][j]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[n][a][4 * j + e] = v[e];
          } else if constexpr (INIT == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[n][a][4 * j + e] = y1[n][a][4 * j + e];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[n][a][4 * j + e] = 0.f;
          }
          *reinterpret_cast<uint2*>(rowp + ((j ^ key) << 4)) = make_uint2(h01, h23);  // LeakyReLU(0) = 0: masked stays 0
        }
      }
    }
    report_f16_saturation(f16_sat16_bad(sat16), p.flags);
  };
  // the accumulators of a conv2 staged in LDS as whole rows
  auto stage_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int n = 0; n < WN; ++n)
#pragma unroll
      for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = wm * 64 + a * 32 + l31;
          *reinterpret_cast<f32x4*>(smem + row * LDO + (2 * wn + n) * 32 + 8 * j + 4 * lh) =
              f32x4{acc[n][a][4 * j], acc[n][a][4 * j + 1], acc[n][a][4 * j + 2], acc[n][a][4 * j + 3]};
        }
  };
  constexpr int V = C / 4, RPP = NTHR / V, NPASS = MT / RPP;  // 32 float4 per row, 8 rows per step, 16 steps
  const int c4 = tid % V, r0 = tid / V;  // = (ct * 8 + cgt, rt) of the patch loads: keep[q] is x at row r0 + 8 q, channels 4 c4 ..
  static_assert(V == 32 && RPP == 8 && NPASS == KEEP, "the output pass must walk the rows the loads kept");

  // ---- first (or only) layer -----------------------------------------------------------------------------------------------------
  conv(prep1, 0, NT1, true);  // its last taps fetch the first taps of conv2
  if constexpr (DIRECT) {  // the residual out of the patch rows' upper pieces, before the buffers are handed over to h
#pragma unroll
    for (int a = 0; a < WM; ++a) {
      int r = arow1[a];
      asm volatile("" : "+v"(r));
      const int row = r + poff1;
      const int key = swz_key(row);
#pragma unroll
      for (int n = 0; n < WN; ++n) {
        const char* rowp = lds + (2 * wn + n) * PBYTES + row * CROW + 8 * lh;
#pragma unroll
        for (int j = 0; j < 4; ++j) rres[n][a][j] = *reinterpret_cast<const u32x2*>(rowp + (((4 + j) << 4) ^ key));
      }
    }
  }
  VFX_TS(5);  // conv1 done
  __syncthreads();  // every wave is done reading the patch buffers that h overlays
  VFX_TS(6);
  if constexpr (DIRECT) write_h(p.b1, hval, std::integral_constant<int, 1>{});
  else write_h(p.b1, hval, std::integral_constant<int, 0>{});
  VFX_TS(7);  // h written
  __syncthreads();  // h is complete
  VFX_TS(8);
  conv(prep2, NT1, 2 * NT1, PAIR);
  VFX_TS(9);  // conv2 done
  if constexpr (!DIRECT || PAIR) __syncthreads();  // every wave is done with h
  VFX_TS(10);
  if constexpr (!DIRECT) {
    stage_acc();
    __syncthreads();
  }
  VFX_TS(11);  // staged

  if constexpr (PAIR && DIRECT) {
    // ---- between the layers, DIRECT: y1 = conv2 (on top of x) + b2 stays in registers in accumulator layout; LeakyReLU(y1) becomes the
    // second layer's operand rows (index m at patch row m + d2; zero outside the sequence and on the two indices of the tile where y1
    // is not valid) -----------------------------------------------------------------------------------------------------------------
    {
      unsigned f16_sat = 0;
#pragma unroll
      for (int n = 0; n < WN; ++n) {
        const int ch = 2 * wn + n;
        f32x4 bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = *(const VFX_GLOBAL f32x4*)(p.b2 + ch * 32 + 8 * j + 4 * lh);
#pragma unroll
        for (int a = 0; a < WM; ++a) {
          const int m = wm * 64 + a * 32 + l31;
          const bool ok = (m >= 1) & (m <= MT - 2) & ((unsigned)(base_h + m) < (unsigned)Tb);
          const int row = m + d2;
          char* rowp = lds + ch * PBYTES + row * CROW + 8 * lh;
          const int key = swz_key(row);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float t = acc[n][a][4 * j + e] + bv[j][e];
              y1[n][a][4 * j + e] = t;
              acc[n][a][4 * j + e] = 0.f;
              v[e] = ok ? fmaxf(t, t * slope) : 0.f;
            }
            *reinterpret_cast<uint2*>(rowp + ((j << 4) ^ key)) = make_uint2(pack_f16x2(v[0], v[1], f16_sat), pack_f16x2(v[2], v[3], f16_sat));
          }
        }
      }
      report_f16_saturation(f16_sat_bits_bad(f16_sat), p.flags);
    }
    __syncthreads();  // the second layer's operand rows are visible
    bool hval2[WM];
    int l31o = l31;
    asm volatile("" : "+v"(l31o));
#pragma unroll
    for (int a = 0; a < WM; ++a) {
      const int m = wm * 64 + a * 32 + l31o;
      hval2[a] = (m >= 1 + d2) & (m <= MT - 2 - d2) & ((unsigned)(base_h + m) < (unsigned)Tb);
    }
    conv(prep1, 2 * NT1, 3 * NT1, true);
    __syncthreads();
    write_h(p.b1b, hval2, std::integral_constant<int, 2>{});
    __syncthreads();
    conv(prep2, 3 * NT1, 4 * NT1, false);
  }
  if constexpr (PAIR && !DIRECT) {
    // ---- between the layers: y1 = conv2 + b2 + x replaces x in the residual registers; LeakyReLU(y1) becomes the operand patch ----
    // (index m at patch row m + d2; zero outside the sequence -- the second layer's zero padding -- and on the two indices of the
    // tile where y1 is not valid).  Every staged row is read before any operand row is written: the two images overlap.
    {
      const f32x4 bv = *(const VFX_GLOBAL f32x4*)(p.b2 + 4 * c4);
#pragma unroll
      for (int q = 0; q < NPASS; ++q) keep[q] = *reinterpret_cast<const f32x4*>(smem + (r0 + q * RPP) * LDO + 4 * c4) + bv + keep[q];
    }
    __syncthreads();
    {
      unsigned f16_sat = 0;
#pragma unroll
      for (int q = 0; q < NPASS; ++q) {
        const int m = r0 + q * RPP;
        const bool ok = (m >= 1) & (m <= MT - 2) & ((unsigned)(base_h + m) < (unsigned)Tb);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ok ? fmaxf(keep[q][e], keep[q][e] * slope) : 0.f;
        const int row = m + d2;
        const int key = (row >> 1) & 7;
        *reinterpret_cast<uint2*>(lds + ct * PBYTES + row * CROW + (((cgt >> 1) ^ key) << 4) + 8 * (cgt & 1)) =
            make_uint2(pack_f16x2(v[0], v[1], f16_sat), pack_f16x2(v[2], v[3], f16_sat));
      }
      report_f16_saturation(f16_sat_bits_bad(f16_sat), p.flags);
    }
    __syncthreads();  // the second layer's operand rows are visible

    // ---- second layer ---------------------------------------------------------------------------------------------------------------
    bool hval2[WM];
    int l31o = l31;
    asm volatile("" : "+v"(l31o));
#pragma unroll
    for (int a = 0; a < WM; ++a) {
      const int m = wm * 64 + a * 32 + l31o;
      hval2[a] = (m >= 1 + d2) & (m <= MT - 2 - d2) & ((unsigned)(base_h + m) < (unsigned)Tb);
    }
    // (cleared here, not when they were staged: 64 live zeros beside the residual and the ring do not fit the register file)
#pragma unroll
    for (int n = 0; n < WN; ++n)
#pragma unroll
      for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][a][r] = 0.f;
    conv(prep1, 2 * NT1, 3 * NT1, true);
    __syncthreads();
    write_h(p.b1b, hval2, std::integral_constant<int, 0>{});
    __syncthreads();
    conv(prep2, 3 * NT1, 4 * NT1, false);
    __syncthreads();
    stage_acc();
    __syncthreads();
  }

  if constexpr (DIRECT) {
    // ---- y = acc + b2 (the residual is inside the accumulators) straight to memory -----------------------------------------------------
    const float* const b2p = PAIR ? p.b2b : p.b2;
    const float aslope = p.act_slope;
    unsigned sat16 = 0;
    constexpr unsigned kOob = 0x80000000u;  // beyond every descriptor of a clip (< 2^28 bytes), + 256 B does not wrap
    // descriptors of the tile's CLIP (round 6): positions past its own end are out of range -- their stores are dropped without a compare
    const int ybytes = Tb * (C * 2);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        p.y ? reinterpret_cast<char*>(p.y) + (int64_t)img * T * (C * 2) : nullptr, 0, p.y ? ybytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rya = __builtin_amdgcn_make_buffer_rsrc(
        p.ya ? reinterpret_cast<char*>(p.ya) + (int64_t)img * T * (C * 2) : nullptr, 0, p.ya ? ybytes : 0, 0x00020000);
    const bool have_y = p.y != nullptr, have_ya = p.ya != nullptr;
    int l31e = l31, lhe = lh;
    asm volatile("" : "+v"(l31e), "+v"(lhe));  // the epilogue's index math stays behind the last conv2
#ifndef VFX_R128_DIRECT_STORES
    // Round 6: y leaves in FULL 128-byte lines.  Stored straight from the MFMA layout a 16-byte store instruction touches 32 rows with
    // 32 bytes each (it writes at 0.65-0.73 of the full-line rate, profiles/r06_c50_direct_store_cost.txt, and the epilogue of this
    // kernel is store-issue-bound: 5.7 k of a block's 27 k cycles).  A wave owns 64 couts = one whole line of each of its 64 positions:
    // it turns its tile through a PRIVATE 9 KB slice of the dead h / patch buffers (144-byte rows: conflict-free column writes) -- one
    // block barrier (every wave is done reading h), no second one.
    constexpr int SROW = 144;
    const int sbase = wave_u * (64 * SROW);
    if (have_y) __syncthreads();
#endif
#pragma unroll
    for (int n = 0; n < WN; ++n) {
      const int ch = 2 * wn + n;
      f32x4 bv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = *(const VFX_GLOBAL f32x4*)(b2p + ch * 32 + 8 * j + 4 * lhe);
#pragma unroll
      for (int a = 0; a < WM; ++a) {
        const int m = wm * 64 + a * 32 + l31e;
        const int li = (int)(((unsigned)m * inv_w1) >> 20), lj = m - li * W1;
        const int pos = PAIR ? base_h + m : base_h + li * rowstride + lj;
        const bool ok = PAIR ? ((m >= 2 + d2) & (m <= MT - 3 - d2))
                             : ((li < TH) & (lj >= 1) & (lj <= W1 - 2) & (!p.fold | (j0 + lj - 1 < d)));
        const unsigned rowoff = ok ? (unsigned)pos * (unsigned)(C * 2) + (unsigned)(ch * 64 + 16 * lhe) : kOob;  // (ok: pos >= 0)
#pragma unroll
        for (int jp = 0; jp < 4; jp += 2) {
          f32x4 v[2];
#pragma unroll
          for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[r][e] = acc[n][a][4 * (jp + r) + e] + bv[jp + r][e];
          // lanes 0-31 keep their run jp and receive the partner's run jp; lanes 32-63 receive the partner's run jp + 1 and keep theirs
          if (have_y) {
            const auto s0 = __builtin_amdgcn_permlane32_swap(pack_f16x2_sat16(v[0][0], v[0][1], true, sat16), pack_f16x2_sat16(v[1][0], v[1][1], true, sat16), false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(pack_f16x2_sat16(v[0][2], v[0][3], true, sat16), pack_f16x2_sat16(v[1][2], v[1][3], true, sat16), false, false);
            const u32x4 w = {s0[0], s1[0], s0[1], s1[1]};
#ifdef VFX_ABL_NOSTORE  // timing-only build (wrong results): what the direct stores cost
            asm volatile("" : : "v"(w));
#elif defined(VFX_R128_DIRECT_STORES)
            __builtin_amdgcn_raw_buffer_store_b128(w, ry, (int)(rowoff + (unsigned)(16 * jp)), 0, 0);
#else
            *reinterpret_cast<u32x4*>(lds + sbase + (a * 32 + l31e) * SROW + n * 64 + 16 * lhe + 16 * jp) = w;
#endif
          }
          if (have_ya) {  // last layer of the stack: also the activated fp16 form for the upsampler that follows
            unsigned q2[2][2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
              f32x4 u;
#pragma unroll
              for (int e = 0; e < 4; ++e) u[e] = fmaxf(v[r][e], v[r][e] * aslope);
              q2[r][0] = pack_f16x2_sat16(u[0], u[1], true, sat16);
              q2[r][1] = pack_f16x2_sat16(u[2], u[3], true, sat16);
            }
            const auto s0 = __builtin_amdgcn_permlane32_swap(q2[0][0], q2[1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(q2[0][1], q2[1][1], false, false);
            const u32x4 w = {s0[0], s1[0], s0[1], s1[1]};
#ifdef VFX_ABL_NOSTORE  // timing-only build (wrong results): what the direct stores cost
            asm volatile("" : : "v"(w));
#else
            __builtin_amdgcn_raw_buffer_store_b128(w, rya, (int)(rowoff + (unsigned)(16 * jp)), 0, 0);
#endif
          }
        }
      }
    }
#if !defined(VFX_R128_DIRECT_STORES) && !defined(VFX_ABL_NOSTORE)
    if (have_y) {
      const int srow = lane >> 3, piece = lane & 7;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int row = srow + 8 * q;  // this wave's position `row` of 64
        const u32x4 w = *reinterpret_cast<const u32x4*>(lds + sbase + row * SROW + 16 * piece);
        const int m = wm * 64 + row;
        const int li = (int)(((unsigned)m * inv_w1) >> 20), lj = m - li * W1;
        const int pos = PAIR ? base_h + m : base_h + li * rowstride + lj;
        const bool ok = PAIR ? ((m >= 2 + d2) & (m <= MT - 3 - d2))
                             : ((li < TH) & (lj >= 1) & (lj <= W1 - 2) & (!p.fold | (j0 + lj - 1 < d)));
        const unsigned off = ok ? (unsigned)pos * (unsigned)(C * 2) + (unsigned)(wn * 128 + 16 * piece) : kOob;
        __builtin_amdgcn_raw_buffer_store_b128(w, ry, (int)off, 0, 0);
      }
    }
#endif
    report_f16_saturation(f16_sat16_bad(sat16), p.flags);
  } else
  // ---- y = conv2 + b2 + residual: whole staged rows read back, the kept rows (x; a pair: y1) added, stored --------------------------
  {
    const f32x4 bv = *(const VFX_GLOBAL f32x4*)((PAIR ? p.b2b : p.b2) + 4 * c4);
    const float aslope = p.act_slope;
    const bool even = (tid & 1) == 0;
    unsigned ya_sat = 0;
    // y and ya through buffer descriptors: 32-bit offsets, and a masked row is an offset beyond the bound (its stores are dropped)
    // -- no exec-mask juggling around 16 (+ 16) stores per thread
    constexpr unsigned kOob = 0xfffffff0u;  // beyond every descriptor (plan_resblock: tensors < 4 GiB - 4096); nothing is added to it
    const unsigned ybytes = (unsigned)((int64_t)p.B * T * C * 4);
    // X16: y is an fp16 tensor like ya; NULL (the last layer in front of an upsampler: only ya is read) = an empty descriptor
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y ? (int)(X16 ? ybytes / 2 : ybytes) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rya = __builtin_amdgcn_make_buffer_rsrc(p.ya, 0, p.ya ? (int)(ybytes / 2) : 0, 0x00020000);
    int r0o = r0;
    if constexpr (PAIR) asm volatile("" : "+v"(r0o));  // recomputed here: row values shared with the pass between the layers would be kept (spilled) through the second layer
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
      const int m = r0o + q * RPP;  // h pixel of the staged row
      const int li = (int)(((unsigned)m * inv_w1) >> 20), lj = m - li * W1;
      const int pos = base_h + li * rowstride + lj;
      const bool ok = PAIR ? ((m >= 2 + d2) & (m <= MT - 3 - d2) & ((unsigned)pos < (unsigned)Tb))
                           : ((li < TH) & (lj >= 1) & (lj <= W1 - 2) & ((unsigned)pos < (unsigned)Tb) & (!p.fold | (j0 + lj - 1 < d)));
      const f32x4 val = *reinterpret_cast<const f32x4*>(smem + m * LDO + 4 * c4) + bv + keep[q];  // + the residual: this thread's own rows
      const unsigned off = (unsigned)(img * T + pos) * (unsigned)(C * 4) + 16u * c4;
      if constexpr (X16) {
        const u32x2 w = {pack_f16x2(val[0], val[1], ya_sat), pack_f16x2(val[2], val[3], ya_sat)};
        __builtin_amdgcn_raw_buffer_store_b64(w, ry, (int)(ok ? off / 2 : kOob), 0, 0);
      } else {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), ry, (int)(ok ? off : kOob), 0, 0);
      }
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
        __builtin_amdgcn_raw_buffer_store_b128(w, rya, (int)((ok && even) ? off / 2 : kOob), 0, 0);
      }
    }
    if (X16 || p.ya) report_f16_saturation(f16_sat_bits_bad(ya_sat), p.flags);
  }
  VFX_TS(12);  // stores issued
  VFX_TS_FLUSH(p.timing, tile, wave_u, NW);
}

int resblock_r128_patch_rows() { return R128_PR; }

// Layer pairs at C = 128: (d, d2) = (1, 3) -- 118 of a tile's 128 indices are outputs; (9, 27) would be 70.
bool resblock_r128_pair_ok(int C, int dil, int dil2, int tuning) {
  return !(tuning & VFX_TUNE_NO_PAIRS) && C == 128 && dil >= 1 && dil <= 16 && dil2 >= 1 && dil2 <= 4;
}

void launch_resblock_r128(const ResBlockParams& hp, const ResBlockParams* dparams, hipStream_t stream) {
  VFX_CHECK(!hp.asrc && hp.hionly && hp.C == 128 && !hp.geo2d && hp.tile_m == 128 && hp.patch_rows == R128_PR,
            "resblock_r128: needs the 16-bit mode, C = 128, 128-position tiles planned with %d patch rows", R128_PR);
  VFX_CHECK(hp.x && (hp.y || (hp.x16 && hp.ya)), "resblock_r128: no input / no output");
  VFX_CHECK(hp.dil2 == 0 || (!hp.fold && hp.w1b && hp.w2b && hp.b1b && hp.b2b && 128 + 2 * hp.dil2 <= R128_PR), "resblock_r128: bad layer pair");
  VFX_CHECK((hp.fold ? hp.PW : hp.dil) <= 32, "resblock_r128: the residual window starts beyond patch row 32");
  const int64_t grid = (int64_t)hp.B * hp.tiles_h * hp.tiles_w;
  VFX_CHECK(grid > 0 && grid < ((int64_t)1 << 31), "resblock_r128: bad grid");
  // 4 chunk buffers of 160 rows = 80 KB (h: 64 KB and the staged tile: 66 KB overlay them): exactly two blocks per CU
  const size_t lds = (size_t)(128 / 32) * R128_PR * CROW;
  static_assert((128 / 32) * R128_PR * CROW >= 128 * (128 + 4) * 4 && (128 / 32) * R128_PR * CROW >= 128 * 128 * 4, "overlays must fit");
  static uint64_t attr_devices = 0;
  if (first_use_on_current_device(attr_devices)) {
    VFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_resblock_r128<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    VFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_resblock_r128<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    VFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_resblock_r128<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    VFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_resblock_r128<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  const dim3 g((int)grid), b(256);
  if (hp.dil2 > 0) {
    if (hp.x16) hipLaunchKernelGGL((k_resblock_r128<true, true>), g, b, lds, stream, dparams);
    else hipLaunchKernelGGL((k_resblock_r128<true, false>), g, b, lds, stream, dparams);
  } else {
    if (hp.x16) hipLaunchKernelGGL((k_resblock_r128<false, true>), g, b, lds, stream, dparams);
    else hipLaunchKernelGGL((k_resblock_r128<false, false>), g, b, lds, stream, dparams);
  }
  VFX_HIP(hipGetLastError());
}

}  // namespace vfx
