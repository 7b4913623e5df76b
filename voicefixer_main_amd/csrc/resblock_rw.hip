// resblock_rw.hip -- the fused ResStack layer of the 16-bit mode for C = 64 (the 44.1 kHz stack: the longest sequences of the
// vocoder, 8 layers of 1.8 GB in + 1.8 GB out) as a PERSISTENT kernel with REGISTER-RESIDENT weights.
//
//     y = x + conv2(LeakyReLU(conv1(LeakyReLU(x)) + b1)) + b2      conv1: k3, dilation d;  conv2: k3, dilation 1
//
// k_resblock<64, 4> (resblock.hip) spends 14 us on a tile whose MFMAs take 0.7 us: a block requests its patch, waits, computes,
// stores, and three blocks per CU (48 KB of LDS each) are all the overlap there is.  A block cannot prefetch its next patch
// behind its own weight fetches either -- vmcnt retires a wave's loads in order, so the first weight wait of a tile would wait
// for the prefetch.  At C = 64 the weights do not have to be fetched per tile at all: a wave's share of BOTH convolutions
// (32 output channels x 64 inputs x 3 taps x 2 convolutions of fp16) is 96 registers.  So here
//   * one block = NW waves stays on its CU and walks a contiguous range of tiles (MT = 32 NW positions of h each);
//   * every wave loads its weight fragments ONCE and keeps them (96 VGPRs; the kernel is built for 256 per wave);
//   * the raw x patch of tile i+1 is requested into registers (global_load, 40 .. 48 VGPRs) right after tile i's patch has
//     been written to LDS, and lands while tile i is computed and stored: no VMEM wait between those two points;
//   * the patch goes to LDS already as fp16 MFMA operands (LeakyReLU applied, 128-byte rows of 64 channels, swizzled pieces):
//     24 / 40 KB instead of 48 KB of raw rows transformed in place;
//   * the centre rows of the patch ARE the residual: they stay in registers (32 VGPRs) and are added in the epilogue, whose
//     thread -> (row, 4 channels) map is the map of those loads: x is read once, and the sum needs no LDS round trip.
// Tile geometry, weights (pack_conv mode 2: fp16 in the hi fragments of 32-channel chunks), arithmetic and summation order are
// those of k_resblock<64, 4, HI> (same products, same summation order).
// MT = 256 (NW = 8, one block per CU): the halo costs 1.25x instead of 1.5x input bytes.  MT = 128 (NW = 4): two blocks per CU.
//
// The kernel is bandwidth-bound (4.1 .. 4.4 TB/s of mixed reads and writes; k_resblock reaches the same with three blocks per
// CU), so the next step is fewer bytes: PAIR = two consecutive layers of small dilation -- (1, 3), (9, 27) -- in ONE pass (NW = 8).
// The first layer's output y1 never leaves the CU: it is the second layer's residual (the registers that held x) and, activated,
// its patch (LDS).  Both layers work over the same 256-index space of the tile (index m = position base + m); a tile advances by
// 252 - 2 d2 outputs.  Per pair x comes in once and y goes out once: 2.05 .. 2.4 instead of 4.0 .. 4.3 units of the tensor size.
// Registers hold conv1 of the first layer only (48); the other three sets of fragments sit in LDS (72 KB, wave-slice-major:
// conflict-free ds_read_b128) -- with more of them in registers the allocator parks fragments in scratch, and a scratch reload
// queues behind the prefetch like any other load.  LDS of a pair (145 KB): R0 = 40 KB, R1 = 32 KB, the fragments, the biases;
//   first layer:  patch R0, h R1;  second layer: patch (= activated y1) R1, h R0;
//   the accumulators are staged for the epilogue in two halves of 128 rows (35 KB over R0) so that R1 is free for the second
//   patch while the first epilogue runs.
//
// X16 (round 4, default): x and y are fp16 tensors (the fp16 trunk of the 16-bit mode, vfx_internal.h): the stack is
// bandwidth-bound, and the values of its fp32 trunk were rounded to fp16 before every MFMA anyway.  A thread's row piece is 8
// bytes (the in-flight patch of the next tile: 20 .. 24 registers instead of 40 .. 48), the operand is formed from the packed
// halves, the residual is widened to fp32 once; sums are fp32, rounded once when y is stored (saturation flagged).  A pair's
// intermediate tensor stays fp32 in registers.
#include <type_traits>

#include "conv_common.h"
#include "vfx_internal.h"

namespace vfx {

// HALO: patch rows beyond the tile's MT positions (PR = MT + HALO).  64 = the round-2 geometry; 128 (fp16 trunk, folded layers
// only): folded tiles as TH x (TW + 2) with TW up to 62 -- a 4 x 63 h tile gives 4 x 61 = 244 outputs of 256 positions where
// 14 x 18 (16-wide tiles) gave 224; its patch is 6 x 63 = 378 rows.  One block per CU owns 160 KB of LDS, the larger patch
// region is free; the two extra row pieces per thread are 4 registers on the fp16 trunk (8 on the fp32 one: over 256).
template <int NW, bool PAIR, bool X16, int HALO = 64>
__global__ __launch_bounds__(NW * 64, 2) void k_resblock_rw(const ResBlockParams* __restrict__ pp, int ntiles, int per_block) {
  constexpr int C = 64;
  constexpr int NTHR = NW * 64;
  constexpr int WM = 2;                     // 32-row MFMA blocks per wave: wave = 64 positions x 32 channels
  constexpr int MT = NW * 32;               // h positions per tile
  constexpr int PR = MT + HALO;             // patch rows (plan_resblock)
  constexpr int RQ = NTHR / 16;             // rows per load group: 16 lanes x 16 bytes = one 256-byte row of raw x
  constexpr int NCQ = MT / RQ;              // centre loads per thread (8)
  constexpr int NHQ = HALO / RQ;            // halo loads per thread (2 or 4)
  static_assert(HALO == 64 || (HALO == 128 && X16 && !PAIR), "the wide patch exists for single layers on the fp16 trunk");
  constexpr int ROWB = 128;                 // bytes per LDS row: 64 channels of fp16
  constexpr int R0 = 0, R1 = PR * ROWB;     // the two operand regions: R0 = PR rows, R1 = MT rows
  constexpr int LDO = C + 4;                // staged output row (floats)
  constexpr int NHALF = PAIR ? 2 : 1;       // the accumulators are staged over R0 (+ R1) at once, or in two halves over R0
  static_assert(NCQ == 8 && (MT / NHALF) * LDO * 4 <= (PAIR ? PR : PR + MT) * ROWB, "staging must fit");
  constexpr int WL_OFF = (PR + MT) * ROWB;  // pairs: weight fragments [conv2 A | conv1 B | conv2 B], 24 KB each
  // DIRECT (round 5; single layers on the fp16 trunk): no staged tile.  The raw fp16 centre rows of the patch go to a third LDS
  // region RX (MT rows, the layout of h) beside their operand form; phase 2 reads them back in ACCUMULATOR layout as the initial
  // value of conv2's accumulators, and the epilogue stores y straight from the accumulators (v_permlane32_swap pairs a lane's
  // 8-byte runs with its partner's into 16-byte stores, cf. resblock_w64.hip).  Per tile this removes 64 KB of staging writes and
  // reads, three of the five block barriers and the 32 residual registers; one block per CU owns the CU's LDS, so RX is free.
  // A PAIR has no room for a third region (145 KB with its weight fragments): its raw rows wait in R1, which is idle until the
  // first h is written -- at the price of one barrier between "every wave has read its residual" and the first h write -- and the
  // first layer's output y1 stays in 32 registers in accumulator layout: activated, it is written as the second layer's operand
  // rows; raw, it is the initial value of the second conv2's accumulators.  7 block barriers per tile instead of 13.
  constexpr bool DIRECT = X16;
  constexpr int RX = PAIR ? R1 : WL_OFF;    // DIRECT: raw fp16 rows of the tile's MT positions
  constexpr int BIAS_OFF = PAIR ? WL_OFF + 3 * 24 * 1024 : (DIRECT ? WL_OFF + MT * ROWB : WL_OFF);  // b1 (pairs: b1, b2, second layer's b1, b2; DIRECT: b1, b2): C floats each
  static_assert(!PAIR || NW == 8, "pairs: 256-position tiles");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* const lds = reinterpret_cast<char*>(smem);

  const ResBlockParams& p = *pp;
  const int tid = threadIdx.x;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int wm = wave_u >> 1, wn = wave_u & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int lr = tid >> 4, cg = tid & 15;
  const int T = p.T, d = p.dil, W1 = p.W1, TH = p.TH, PW = p.PW, P = p.P;
  const int rowstride = p.fold ? d : 0;
  const int tiles_per_img = p.tiles_w * p.tiles_h;
  const float slope = p.slope;
  const int d2 = p.dil2;           // PAIR: the second layer's dilation
  const int c0 = p.fold ? PW : d;  // patch row of h pixel 0: patch row m + c0 holds the input sample AT h pixel m (the residual)

  // ---- weights: this wave's 32 output channels, all taps, for the lifetime of the block ---------------------------------
  // registers: [conv][32-channel chunk][tap][K = 16 step] (pairs: conv1 of the first layer only);  LDS (pairs): 1 KB per
  // (set, chunk, tap, K step, 32-channel half), lane * 16 inside
  f16x8 W[PAIR ? 1 : 2][2][3][2];
  {
    const int64_t ts = (int64_t)C * kKC;
    const unsigned nb_off = (unsigned)(wn * 1024 + lane * 4) * 4u;
#pragma unroll
    for (int set = 0; set < (PAIR ? 4 : 2); ++set)  // conv1 A, conv2 A, conv1 B, conv2 B
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float* wt = set == 0 ? p.w1 : (set == 1 ? p.w2 : (set == 2 ? p.w1b : p.w2b));
          const char* w = reinterpret_cast<const char*>(wt + (3 * c + k) * ts) + nb_off;
          if (PAIR && set >= 1) {
            if (wm == 0) {
#pragma unroll
              for (int s2 = 0; s2 < 2; ++s2)
                *reinterpret_cast<f32x4*>(lds + WL_OFF + (((((set - 1) * 6 + 3 * c + k) * 2 + s2) * 2 + wn) << 10) + lane * 16) =
                    *(const VFX_GLOBAL f32x4*)(w + 2048 * s2);
            }
          } else {
            W[set][c][k][0] = __builtin_bit_cast(f16x8, *(const VFX_GLOBAL f32x4*)(w));
            W[set][c][k][1] = __builtin_bit_cast(f16x8, *(const VFX_GLOBAL f32x4*)(w + 2048));
          }
        }
  }
  // conv1's bias lives in LDS (a VMEM load per tile would queue behind the prefetch), conv2's in 4 registers (pairs: all four
  // vectors in LDS)
  float* const b1s = reinterpret_cast<float*>(lds + BIAS_OFF);
  if (tid < C) b1s[tid] = ((const VFX_GLOBAL float*)p.b1)[tid];
  f32x4 b2v = *(const VFX_GLOBAL f32x4*)(p.b2 + 4 * cg);
  if constexpr (DIRECT && !PAIR) {
    if (tid < C) b1s[C + tid] = ((const VFX_GLOBAL float*)p.b2)[tid];
  }
  if constexpr (PAIR) {
    if (tid < C) {
      b1s[C + tid] = ((const VFX_GLOBAL float*)p.b2)[tid];
      b1s[2 * C + tid] = ((const VFX_GLOBAL float*)p.b1b)[tid];
      b1s[3 * C + tid] = ((const VFX_GLOBAL float*)p.b2b)[tid];
    }
  }

  // Per-thread geometry is tile-independent but is RECOMPUTED per tile (exact magic-number divisions, a few VALU operations
  // per row): as tables it costs 40 registers that the 256 of this kernel do not have.
  const unsigned inv_pw = ((1u << 20) + PW - 1) / PW, inv_w1 = ((1u << 20) + W1 - 1) / W1;  // rows < 512, divisors <= 320: exact
  int lr_v = lr, l31_v = l31;  // made opaque per tile so that the compiler does not hoist (and spill) those tables itself
  // sample offset of patch row pr from the patch origin, or -1: no such row
  auto rel_of = [&](int pr) __attribute__((always_inline)) {
    const int pi = (int)(((unsigned)pr * inv_pw) >> 20);
    return pr < P ? pi * rowstride + (pr - pi * PW) : -1;
  };
  // loads: centre q = patch row lr + RQ q + c0 (= h pixel lr + RQ q); halo q = the rows in front of / behind the centre
  auto crow = [&](int q) __attribute__((always_inline)) { return lr_v + RQ * q + c0; };
  auto hrow = [&](int q) __attribute__((always_inline)) {
    const int hr = lr_v + RQ * q;
    return hr < c0 ? hr : hr + MT;
  };
  // sample offset of h pixel m from h pixel (0, 0), far outside every sequence if the tile has no such pixel
  auto hrel_of = [&](int m) __attribute__((always_inline)) {
    const int li = (int)(((unsigned)m * inv_w1) >> 20), lj = m - li * W1;
    return li < TH ? li * rowstride + lj : -(1 << 29);
  };

  const int t_begin = blockIdx.x * per_block, t_end = min(t_begin + per_block, ntiles);
  auto tile_geom = [&](int t, int& img, int& j0, int& base_h) __attribute__((always_inline)) {
    const int tj = t % p.tiles_w;
    const int ti = (t / p.tiles_w) % p.tiles_h;
    img = t / tiles_per_img;
    j0 = tj * p.TWo;
    base_h = PAIR ? j0 - 2 - d2 : (p.fold ? ti * TH * d + j0 - 1 : j0 - 1);  // position of h pixel 0 (pairs: of index 0, below)
  };

  typedef typename std::conditional<X16, u32x2, f32x4>::type ld_t;  // this thread's 4 channels of one row of x
  ld_t PC[NCQ], PH[NHQ];  // the raw patch of the NEXT tile, in flight / landed
  auto request = [&](int t) __attribute__((always_inline)) {
    int img, j0, base_h;
    tile_geom(t, img, j0, base_h);
    const int base_x = base_h - d;
    // the end of THAT tile's clip (ResBlockParams::lens; on the fp16 trunk only -- the fp32-trunk variants of VFX_TUNE_F32_TRUNK have
    // no register left for it and the plan refuses the combination)
    const int Tn = (X16 && p.lens) ? min(T, ((const VFX_GLOBAL int*)p.lens)[__builtin_amdgcn_readfirstlane(img)] * p.lens_mul) : T;
    // (element offsets; X16: fp16 elements)
    const char* xi = reinterpret_cast<const char*>(p.x) + ((int64_t)img * T * C + 4 * cg) * (X16 ? 2 : 4);
#pragma unroll
    for (int q = 0; q < NCQ; ++q) {
      const int rel = rel_of(crow(q)), pos = base_x + rel;
      PC[q] = ld_t{};
      if (rel >= 0 && (unsigned)pos < (unsigned)Tn) PC[q] = *(const VFX_GLOBAL ld_t*)(xi + (int64_t)pos * C * (X16 ? 2 : 4));
    }
#pragma unroll
    for (int q = 0; q < NHQ; ++q) {
      const int rel = rel_of(hrow(q)), pos = base_x + rel;
      PH[q] = ld_t{};
      if (rel >= 0 && (unsigned)pos < (unsigned)Tn) PH[q] = *(const VFX_GLOBAL ld_t*)(xi + (int64_t)pos * C * (X16 ? 2 : 4));
    }
  };
  // raw row -> LeakyReLU -> fp16 -> this thread's 8 bytes of the patch row: piece cg >> 1 (8 channels) at slot piece ^ key
  auto to_patch = [&](char* patch, const f32x4& raw, int pr, unsigned& sat) __attribute__((always_inline)) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(raw[e], raw[e] * slope);
    *reinterpret_cast<uint2*>(patch + pr * ROWB + (((cg >> 1) ^ ((pr >> 1) & 7)) << 4) + 8 * (cg & 1)) =
        make_uint2(pack_f16x2(v[0], v[1], sat), pack_f16x2(v[2], v[3], sat));
  };
  // X16: the row piece is fp16 already -- LeakyReLU on the packed halves (max(x, slope x), 0 < slope < 1), nothing to convert
  auto to_patch16 = [&](char* patch, const u32x2& raw, int pr) __attribute__((always_inline)) {
    const u32x2 v = f16x4_lrelu(raw, slope);
    *reinterpret_cast<uint2*>(patch + pr * ROWB + (((cg >> 1) ^ ((pr >> 1) & 7)) << 4) + 8 * (cg & 1)) = make_uint2(v.x, v.y);
  };

  // batches of clips of unequal length (ResBlockParams::lens): the end of the CURRENT tile's clip, set per tile -- positions past it
  // read as zeros, h (and a pair's intermediate tensor) is zero there, nothing is stored there
  int Tb = T;
  VFX_TS_DECL;  // timing builds (-DVFX_TIMING, scripts/phase_timing.py): per tile and wave, s_memtime at the phase boundaries
  f32x16 y1r[WM];  // DIRECT pairs: the first layer's output in accumulator layout (the second layer's residual)
  f32x16 acc[WM];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  // one tap of one 32-channel chunk: A rows `row[a]` of an LDS image of 128-byte rows
  auto mma = [&](const f16x8 (&w)[2], const char* img_base, const int (&row)[WM], int c) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f16x8 ah[WM];
#pragma unroll
      for (int a = 0; a < WM; ++a)
        ah[a] = *reinterpret_cast<const f16x8*>(img_base + row[a] * ROWB + ((64 * c + 32 * s + 16 * lh) ^ swz_key(row[a])));
#pragma unroll
      for (int a = 0; a < WM; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[s], ah[a], acc[a], 0, 0, 0);
    }
  };
  // weight fragments of set 0 .. 3 = conv1 A, conv2 A, conv1 B, conv2 B (registers, or LDS for the sets a pair keeps there)
  auto mma_set = [&](auto set_tag, int c, int k, const char* img_base, const int (&row)[WM]) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_tag)::value;
    if constexpr (PAIR && SET >= 1) {
      f16x8 w[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
        w[s2] = *reinterpret_cast<const f16x8*>(lds + WL_OFF + (((((SET - 1) * 6 + 3 * c + k) * 2 + s2) * 2 + wn) << 10) + lane * 16);
      mma(w, img_base, row, c);
    } else {
      mma(W[SET][c][k], img_base, row, c);
    }
  };

  // conv1 -> h -> conv2 of ONE layer over the index space of the tile; the accumulators hold conv2 on return.
  //   SECOND = false: A rows of conv1 are patch rows arow1[a] + poff[k] of the x patch (R0), h goes to R1;
  //   SECOND = true (pairs): the patch is the first layer's activated output over the SAME index space (R1; rows
  //   m + (k - 1) d2, clamped: rows that would need an index outside the tile only feed results that are never stored), h to R0.
  auto layer = [&](auto second_tag, const int (&arow1)[WM], int base_h) __attribute__((always_inline)) {
    constexpr bool SECOND = decltype(second_tag)::value;
    const char* patch = lds + (SECOND ? R1 : R0);
    char* hbuf = lds + (SECOND ? R0 : R1);
    const float* bias1 = b1s + (SECOND ? 2 * C : 0);
    // conv1 (chunk-major taps: the order of k_resblock)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        int rows[WM];
#pragma unroll
        for (int a = 0; a < WM; ++a) {
          if constexpr (SECOND) {
            const int r = (wm * WM + a) * 32 + l31_v + (k - 1) * d2;
            rows[a] = r < 0 ? 0 : (r > MT - 1 ? MT - 1 : r);
          } else {
            rows[a] = arow1[a] + p.poff[k];
          }
        }
        mma_set(std::integral_constant<int, SECOND ? 2 : 0>{}, c, k, patch, rows);
      }
    VFX_TS(SECOND ? 9 : 4);  // conv1 done
    // h = LeakyReLU(conv1 + b1) as fp16 operands, zero outside the sequence
    {
      unsigned sat16 = 0;
      const f16x2 slope2 = {(_Float16)slope, (_Float16)slope};
      u32x2 rres[WM][4];  // DIRECT: this lane's residual pieces (pixel m, channels wn * 32 + 8 j + 4 lh .. + 3) out of RX
      if constexpr (DIRECT && !SECOND) {
#pragma unroll
        for (int a = 0; a < WM; ++a) {
          const int m = (wm * WM + a) * 32 + l31_v;
          const char* rowx = lds + RX + m * ROWB + 8 * lh;
          const int key = (m >> 1) & 7;
#pragma unroll
          for (int j = 0; j < 4; ++j) rres[a][j] = *reinterpret_cast<const u32x2*>(rowx + (((wn * 4 + j) ^ key) << 4));
        }
        if constexpr (PAIR) __syncthreads();  // RX = R1 = the h buffer of this layer: every wave has its residual before any h is written
      }
#pragma unroll
      for (int a = 0; a < WM; ++a) {
        const int m = (wm * WM + a) * 32 + l31_v;
        const bool hval = (unsigned)(base_h + (SECOND ? m : hrel_of(m))) < (unsigned)Tb;
        char* rowp = hbuf + m * ROWB + 8 * lh;
        const int key = (m >> 1) & 7;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 b1v = *reinterpret_cast<const f32x4*>(bias1 + wn * 32 + 8 * j + 4 * lh);
          // convert first, activate the packed halves (conv_common.h: pack_f16x2_sat16 / lrelu_f16x2)
          const unsigned h01 = lrelu_f16x2(pack_f16x2_sat16(acc[a][4 * j] + b1v[0], acc[a][4 * j + 1] + b1v[1], hval, sat16), slope2);
          const unsigned h23 = lrelu_f16x2(pack_f16x2_sat16(acc[a][4 * j + 2] + b1v[2], acc[a][4 * j + 3] + b1v[3], hval, sat16), slope2);
          if constexpr (DIRECT && !SECOND) {  // conv2 accumulates on top of the residual
            const f32x4 v = f16x4_widen(rres[a][j]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[a][4 * j + e] = v[e];
          } else if constexpr (DIRECT) {     // a pair's second layer: on top of y1
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[a][4 * j + e] = y1r[a][4 * j + e];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[a][4 * j + e] = 0.f;
          }
          *reinterpret_cast<uint2*>(rowp + (((wn * 4 + j) ^ key) << 4)) = make_uint2(h01, h23);  // LeakyReLU(0) = 0: masked stays 0
        }
      }
      report_f16_saturation(f16_sat16_bad(sat16), p.flags);
    }
    VFX_TS(SECOND ? 10 : 5);  // residual read, h written
    __syncthreads();  // h is complete
    VFX_TS(SECOND ? 11 : 6);
    // conv2 from the resident h
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        int rows[WM];
#pragma unroll
        for (int a = 0; a < WM; ++a) {
          const int r = (wm * WM + a) * 32 + l31_v + k - 1;
          rows[a] = r < 0 ? 0 : (r > MT - 1 ? MT - 1 : r);  // clamped rows only feed outputs that are masked anyway
        }
        mma_set(std::integral_constant<int, SECOND ? 3 : 1>{}, c, k, hbuf, rows);
      }
    // every wave is done with the patch and h: the staged accumulators may overlay them.  (DIRECT stages nothing: the next
    // tile's patch goes to R0 / RX, last read before this tile's "h is complete" barrier, and h is rewritten only behind the
    // next tile's "patch is complete" barrier, which no wave passes before every wave has left this conv2.)
    if constexpr (!DIRECT || PAIR) __syncthreads();
  };
  // the accumulators of rows [half * MT / NHALF, (half + 1) * MT / NHALF) to the staging rows (floats, over R0 (+ R1))
  auto stage = [&](int half) __attribute__((always_inline)) {
    if (NHALF == 2 && (wm >> 1) != half) return;  // wave-uniform: waves wm = 0, 1 hold the first 128 rows
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = (wm * WM + a) * 32 + l31_v - half * (MT / NHALF);
        *reinterpret_cast<f32x4*>(smem + row * LDO + wn * 32 + 8 * j + 4 * lh) =
            f32x4{acc[a][4 * j], acc[a][4 * j + 1], acc[a][4 * j + 2], acc[a][4 * j + 3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[a][4 * j + e] = 0.f;
      }
  };

  if (t_begin < t_end) request(t_begin);
  __syncthreads();  // the biases (and the fragments a pair keeps in LDS) are there
  for (int t = t_begin; t < t_end; ++t) {
    VFX_TS(0);
    asm volatile("" : "+v"(lr_v), "+v"(l31_v));
    int img, j0, base_h;
    tile_geom(t, img, j0, base_h);
    if (X16 && p.lens) {
      Tb = min(T, ((const VFX_GLOBAL int*)p.lens)[__builtin_amdgcn_readfirstlane(img)] * p.lens_mul);
      if ((PAIR ? j0 : base_h + 1) >= Tb) {  // the tile lies wholly past the end of its clip: nothing to compute or store
        if (t + 1 < t_end) request(t + 1);
        continue;
      }
    }
    int arow1[WM];  // patch row of this lane's h pixels
#pragma unroll
    for (int a = 0; a < WM; ++a) {
      const int ml = (wm * WM + a) * 32 + l31_v;
      const int li = (int)(((unsigned)ml * inv_w1) >> 20), lj = ml - li * W1;
      arow1[a] = li < TH ? li * PW + lj : 0;
    }

    // ---- the landed patch: to LDS as operands; its centre stays as the residual --------------------------------------
    f32x4 K[NCQ];
    {
      unsigned sat = 0;
      if constexpr (X16) {
#pragma unroll
        for (int q = 0; q < NCQ; ++q) {
          if constexpr (DIRECT) {  // the raw row piece beside its operand form: h pixel m = lr + RQ q, the layout of h
            const int m = lr_v + RQ * q;
            *reinterpret_cast<uint2*>(lds + RX + m * ROWB + (((cg >> 1) ^ ((m >> 1) & 7)) << 4) + 8 * (cg & 1)) = make_uint2(PC[q].x, PC[q].y);
          } else {
            K[q] = f16x4_widen(PC[q]);
          }
          if (crow(q) < P) to_patch16(lds + R0, PC[q], crow(q));
        }
#pragma unroll
        for (int q = 0; q < NHQ; ++q)
          if (hrow(q) < P) to_patch16(lds + R0, PH[q], hrow(q));
      } else {
#pragma unroll
        for (int q = 0; q < NCQ; ++q) {
          K[q] = PC[q];
          if (crow(q) < P) to_patch(lds + R0, PC[q], crow(q), sat);
        }
#pragma unroll
        for (int q = 0; q < NHQ; ++q)
          if (hrow(q) < P) to_patch(lds + R0, PH[q], hrow(q), sat);
        report_f16_saturation(f16_sat_bits_bad(sat), p.flags);
      }
    }
    VFX_TS(1);  // patch rows written
    __syncthreads();  // the patch is complete
    VFX_TS(2);
    if (t + 1 < t_end) request(t + 1);  // lands while this tile is computed and stored
    VFX_TS(3);  // next patch requested

    layer(std::false_type{}, arow1, base_h);
    VFX_TS(7);  // first (only) layer: conv2 done

    if constexpr (PAIR && DIRECT) {
      // ---- between the layers, DIRECT: y1 = acc (conv2 on top of x) + b2 stays in registers; LeakyReLU(y1) becomes the second layer's
      // operand rows in R1 (index m at row m; zero outside the sequence and on the two indices where y1 is not valid) -------------------
      unsigned sat = 0;
#pragma unroll
      for (int a = 0; a < WM; ++a) {
        const int m = (wm * WM + a) * 32 + l31_v;
        const bool ok = (m >= 1) & (m <= MT - 2) & ((unsigned)(base_h + m) < (unsigned)Tb);
        char* rowp = lds + R1 + m * ROWB + 8 * lh;
        const int key = (m >> 1) & 7;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 b2a = *reinterpret_cast<const f32x4*>(b1s + C + wn * 32 + 8 * j + 4 * lh);
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float t = acc[a][4 * j + e] + b2a[e];
            y1r[a][4 * j + e] = t;
            acc[a][4 * j + e] = 0.f;
            v[e] = ok ? fmaxf(t, t * slope) : 0.f;
          }
          *reinterpret_cast<uint2*>(rowp + (((wn * 4 + j) ^ key) << 4)) = make_uint2(pack_f16x2(v[0], v[1], sat), pack_f16x2(v[2], v[3], sat));
        }
      }
      report_f16_saturation(f16_sat_bits_bad(sat), p.flags);
      __syncthreads();  // the second patch is complete
      VFX_TS(8);  // y1 formed, second patch written, barrier passed
      layer(std::true_type{}, arow1, base_h);
      VFX_TS(12);  // second layer: conv2 done
    }
    if constexpr (PAIR && !DIRECT) {
      // ---- first layer's epilogue: y1 = conv2 + x + b2 stays on the CU -- as the second layer's residual (registers, same
      // thread -> row map) and, activated, as its patch (R1).  Index m of the tile = position base_h + m for BOTH layers: the
      // first layer's outputs are m = 1 .. MT-2, the second one's h is right for m = 1+d2 .. MT-2-d2, its outputs for
      // m = 2+d2 .. MT-3-d2 (the MT - 4 - 2 d2 positions a tile advances by).  Outside the sequence y1 is zero (the padding
      // of the second layer's conv1).
      unsigned sat = 0;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        stage(half);
        __syncthreads();
#pragma unroll
        for (int qq = 0; qq < NCQ / 2; ++qq) {
          const int q = half * (NCQ / 2) + qq;
          const int m = lr_v + RQ * q;
          const bool ok = (m >= 1) & (m <= MT - 2) & ((unsigned)(base_h + m) < (unsigned)Tb);
          const f32x4 b2a = *reinterpret_cast<const f32x4*>(b1s + C + 4 * cg);
          const f32x4 val = (*reinterpret_cast<const f32x4*>(smem + (m - half * (MT / 2)) * LDO + 4 * cg) + K[q]) + b2a;
          K[q] = ok ? val : f32x4{0.f, 0.f, 0.f, 0.f};
          to_patch(lds + R1, K[q], m, sat);
        }
        __syncthreads();  // the staged half has been read (second half: the second patch is complete)
      }
      report_f16_saturation(f16_sat_bits_bad(sat), p.flags);
      layer(std::true_type{}, arow1, base_h);
      b2v = *reinterpret_cast<const f32x4*>(b1s + 3 * C + 4 * cg);
    }

    if constexpr (DIRECT) {
      // ---- epilogue, DIRECT: y = acc + b2 (the residual is inside the accumulators) straight to memory -----------------------------
      char* const yb = reinterpret_cast<char*>(p.y) + ((int64_t)img * T * C + wn * 32 + 8 * lh) * 2;
      char* const yab = reinterpret_cast<char*>(p.ya) + ((int64_t)img * T * C + wn * 32 + 8 * lh) * 2;
      const bool have_y = p.y != nullptr, have_ya = p.ya != nullptr;
      const float aslope = p.act_slope;
      unsigned sat = 0;
      f32x4 b2r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b2r[j] = *reinterpret_cast<const f32x4*>(b1s + (PAIR ? 3 * C : C) + wn * 32 + 8 * j + 4 * lh);
#pragma unroll
      for (int a = 0; a < WM; ++a) {
        const int m = (wm * WM + a) * 32 + l31_v;
        const int li = (int)(((unsigned)m * inv_w1) >> 20), lj = m - li * W1;
        const int pos = PAIR ? base_h + m : base_h + li * rowstride + lj;
        const bool ok = PAIR ? ((m >= 2 + d2) & (m <= MT - 3 - d2) & ((unsigned)pos < (unsigned)Tb))
                             : ((li < TH) & (lj >= 1) & (lj <= W1 - 2) & ((unsigned)pos < (unsigned)Tb) & (!p.fold | (j0 + lj - 1 < d)));
#pragma unroll
        for (int jp = 0; jp < 4; jp += 2) {
          f32x4 v[2];
#pragma unroll
          for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[r][e] = acc[a][4 * (jp + r) + e] + b2r[jp + r][e];
              acc[a][4 * (jp + r) + e] = 0.f;
            }
          // lanes 0-31 keep their run jp and receive the partner's run jp; lanes 32-63 receive the partner's run jp + 1 and keep theirs
          if (have_y) {
            const auto s0 = __builtin_amdgcn_permlane32_swap(pack_f16x2(v[0][0], v[0][1], sat), pack_f16x2(v[1][0], v[1][1], sat), false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(pack_f16x2(v[0][2], v[0][3], sat), pack_f16x2(v[1][2], v[1][3], sat), false, false);
            const u32x4 w = {s0[0], s1[0], s0[1], s1[1]};
            if (ok) *(VFX_GLOBAL u32x4*)(yb + (int64_t)pos * C * 2 + 16 * jp) = w;
          }
          if (have_ya) {  // last layer in front of an upsampler: ya = fp16(LeakyReLU(y, act_slope))
            unsigned q2[2][2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
              f32x4 u;
#pragma unroll
              for (int e = 0; e < 4; ++e) u[e] = fmaxf(v[r][e], v[r][e] * aslope);
              q2[r][0] = pack_f16x2(u[0], u[1], sat);
              q2[r][1] = pack_f16x2(u[2], u[3], sat);
            }
            const auto s0 = __builtin_amdgcn_permlane32_swap(q2[0][0], q2[1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(q2[0][1], q2[1][1], false, false);
            const u32x4 w = {s0[0], s1[0], s0[1], s1[1]};
            if (ok) *(VFX_GLOBAL u32x4*)(yab + (int64_t)pos * C * 2 + 16 * jp) = w;
          }
        }
      }
      report_f16_saturation(f16_sat_bits_bad(sat), p.flags);
      VFX_TS(13);  // stores issued
      VFX_TS_FLUSH(p.timing, t, wave_u, NW);
    } else
    // ---- epilogue: y = conv2 + residual + b2 in the layout of the centre loads ----------------------------------------------
    {
      char* const yi = reinterpret_cast<char*>(p.y) + ((int64_t)img * T * C + 4 * cg) * (X16 ? 2 : 4);
      const bool have_y = p.y != nullptr;  // X16: NULL when only ya is consumed
      const bool even = (tid & 1) == 0;
      const float aslope = p.act_slope;
      unsigned sat = 0;
#pragma unroll
      for (int half = 0; half < NHALF; ++half) {
        stage(half);
        __syncthreads();
#pragma unroll
        for (int qq = 0; qq < NCQ / NHALF; ++qq) {
          const int q = half * (NCQ / NHALF) + qq;
          const int m = lr_v + RQ * q;
          int pos;
          bool ok;
          if constexpr (PAIR) {
            pos = base_h + m;
            ok = (m >= 2 + d2) & (m <= MT - 3 - d2) & ((unsigned)pos < (unsigned)Tb);
          } else {
            const int li = (int)(((unsigned)m * inv_w1) >> 20), lj = m - li * W1;
            pos = base_h + li * rowstride + lj;
            ok = (li < TH) & (lj >= 1) & (lj <= W1 - 2) & ((unsigned)pos < (unsigned)Tb) & (!p.fold | (j0 + lj - 1 < d));
          }
          const f32x4 val = (*reinterpret_cast<const f32x4*>(smem + (m - half * (MT / NHALF)) * LDO + 4 * cg) + K[q]) + b2v;
          if constexpr (X16) {
            const u32x2 w16 = {pack_f16x2(val[0], val[1], sat), pack_f16x2(val[2], val[3], sat)};
            if (ok && have_y) *(VFX_GLOBAL u32x2*)(yi + (int64_t)pos * C * 2) = w16;
          } else {
            if (ok) *(VFX_GLOBAL f32x4*)(yi + (int64_t)pos * C * 4) = val;
          }
          if (p.ya) {  // last layer in front of an upsampler: also ya = fp16(LeakyReLU(y, act_slope)), cf. k_resblock
            f32x4 u;
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = fmaxf(val[e], val[e] * aslope);
            const unsigned h01 = pack_f16x2(u[0], u[1], sat), h23 = pack_f16x2(u[2], u[3], sat);
            const unsigned g0 = (unsigned)__builtin_amdgcn_mov_dpp((int)h01, 0xB1, 0xf, 0xf, false);
            const unsigned g1 = (unsigned)__builtin_amdgcn_mov_dpp((int)h23, 0xB1, 0xf, 0xf, false);
            const u32x4 w = {h01, h23, g0, g1};
            if (ok && even) *(VFX_GLOBAL f32x4*)(p.ya + ((int64_t)img * T + pos) * (C / 2) + 2 * cg) = __builtin_bit_cast(f32x4, w);
          }
        }
        __syncthreads();  // the staged rows have been read: the next half / the next patch may overwrite them
      }
      if (X16 || p.ya) report_f16_saturation(f16_sat_bits_bad(sat), p.flags);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// k_resblock_rw16 (round 6): the same layer / layer pair on the fp16 trunk (X16, DIRECT) with a tile loop that spends its
// instructions on the tile.  The phase stamps of round 6 (profiles/r06_c2_phase_timing_c64_before.txt) and the instruction counts of
// the loop body say what k_resblock_rw<.., X16> is bound by: ~1 830 instructions per tile and wave, 48 of them MFMAs, at ~8 cycles
// each -- three integer divisions per tile_geom() (twice per tile), 22 instructions per 8-byte row piece of the patch request (a
// multiply-shift division, two exec-mask save / restore pairs, a branch, 64-bit address arithmetic), twelve such pieces per
// thread, exec-masked stores, the tile's geometry recomputed three times.  Two 4-wave blocks per CU instead of one 8-wave block
// (profiles/r06_c2_voc_layers_rw_4wave_blocks_loser.txt) changed nothing: the waves are issue-bound, not latency-bound.  Here:
//   * a thread's row piece is 16 bytes (8 channels; 8 lanes per 128-byte row): 4 + 1 (+ 1) loads per tile instead of 8 + 2 (+ 2),
//     one ds_write_b128 per operand piece and one per raw piece;
//   * the loads and stores go through buffer descriptors that cover exactly the tile's CLIP (base = clip, num_records = its own
//     length): positions in front of / behind the clip are out of range -- zeros on loads, dropped stores -- with no compare, no
//     exec mask and no branch; a load is v_add + buffer_load, its byte offset inside the patch a per-thread constant;
//   * everything that depends on the lane only (patch row of an h pixel, its position offset, its output offset, the LDS
//     addresses of the patch writes) is computed once per block;
//   * the tile coordinates advance by increments (no division after the first tile);
//   * a varlen batch reads the clip's length with a scalar load (the global load the compiler made of it waited vmcnt(0): for
//     every store of the previous tile).
// Same products, same sums, same rounding as k_resblock_rw<8, PAIR, true, HALO>: bit-identical results.
// Measured and NOT kept (profiles/r06_c4_voc_layers_store_pairing_and_nt_losers.txt, same box, alternating): pairing the two 32-byte
// pieces a lane pair holds of a row with those of the pixel 16 lanes away (v_permlane16_swap: two stores of 64 bytes of 16 rows each
// instead of two of 32 bytes of all 32 rows) -- singles 2.34-2.38 -> 2.85 ms; the same with non-temporal stores (aux = 2): 2.95 ms.
template <bool PAIR, int HALO = 64, bool DOWN = false>
__global__ __launch_bounds__(512, 2) void k_resblock_rw16(const ResBlockParams* __restrict__ pp, int ntiles, int per_block) {
  static_assert(!(PAIR && DOWN), "pairs have no folded tiles");
  constexpr int C = 64, NW = 8, NTHR = NW * 64, WM = 2, MT = 256, PR = MT + HALO;
  constexpr int RQ = NTHR / 8;              // rows per load group: 8 lanes x 16 bytes = one 128-byte row of the fp16 trunk
  constexpr int NCQ = MT / RQ;              // centre loads per thread (4)
  constexpr int NHQ = HALO / RQ;            // halo loads per thread (1 or 2)
  constexpr int ROWB = 128;
  constexpr int R0 = 0, R1 = PR * ROWB;
  constexpr int WL_OFF = (PR + MT) * ROWB;
  constexpr int RX = PAIR ? R1 : WL_OFF;    // raw fp16 rows of the tile's MT positions (a pair parks them in the idle h region)
  constexpr int BIAS_OFF = PAIR ? WL_OFF + 3 * 24 * 1024 : WL_OFF + MT * ROWB;
  constexpr unsigned kOob = 0x80000000u;    // beyond every descriptor of a clip (< 2^28 bytes), also after a tile's base is added
  static_assert(HALO == 64 || (HALO == 128 && !PAIR), "the wide patch exists for single layers");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* const lds = reinterpret_cast<char*>(smem);

  const ResBlockParams& p = *pp;
  const int tid = threadIdx.x;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int wm = wave_u >> 1, wn = wave_u & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int lr8 = tid >> 3, c8 = tid & 7;
  const int T = p.T, d = p.dil, W1 = p.W1, TH = p.TH, PW = p.PW, P = p.P;
  const int rowstride = p.fold ? d : 0;
  const int tiles_w = p.tiles_w, tiles_h = p.tiles_h, TWo = p.TWo;
  const float slope = p.slope;
  const int d2 = p.dil2;
  const int c0 = p.fold ? PW : d;  // patch row of h pixel 0

  // ---- weights (as k_resblock_rw): this wave's 32 output channels, all taps, for the lifetime of the block ------------------------
  f16x8 W[PAIR ? 1 : 2][2][3][2];
  {
    const int64_t ts = (int64_t)C * kKC;
    const unsigned nb_off = (unsigned)(wn * 1024 + lane * 4) * 4u;
#pragma unroll
    for (int set = 0; set < (PAIR ? 4 : 2); ++set)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float* wt = set == 0 ? p.w1 : (set == 1 ? p.w2 : (set == 2 ? p.w1b : p.w2b));
          const char* w = reinterpret_cast<const char*>(wt + (3 * c + k) * ts) + nb_off;
          if (PAIR && set >= 1) {
            if (wm == 0) {
#pragma unroll
              for (int s2 = 0; s2 < 2; ++s2)
                *reinterpret_cast<f32x4*>(lds + WL_OFF + (((((set - 1) * 6 + 3 * c + k) * 2 + s2) * 2 + wn) << 10) + lane * 16) =
                    *(const VFX_GLOBAL f32x4*)(w + 2048 * s2);
            }
          } else {
            W[set][c][k][0] = __builtin_bit_cast(f16x8, *(const VFX_GLOBAL f32x4*)(w));
            W[set][c][k][1] = __builtin_bit_cast(f16x8, *(const VFX_GLOBAL f32x4*)(w + 2048));
          }
        }
  }
  float* const b1s = reinterpret_cast<float*>(lds + BIAS_OFF);  // b1, b2 (pairs: + the second layer's b1, b2): C floats each
  if (tid < C) {
    b1s[tid] = ((const VFX_GLOBAL float*)p.b1)[tid];
    b1s[C + tid] = ((const VFX_GLOBAL float*)p.b2)[tid];
    if constexpr (PAIR) {
      b1s[2 * C + tid] = ((const VFX_GLOBAL float*)p.b1b)[tid];
      b1s[3 * C + tid] = ((const VFX_GLOBAL float*)p.b2b)[tid];
    }
  }

  // ---- per-lane geometry, once per block -------------------------------------------------------------------------------------------
  const unsigned inv_pw = ((1u << 20) + PW - 1) / PW, inv_w1 = ((1u << 20) + W1 - 1) / W1;  // rows < 512, divisors <= 320: exact
  int arow1[WM];   // patch row of this lane's h pixel m = (wm WM + a) 32 + l31
  int hrel[WM];    // its position relative to h pixel (0, 0); far negative when the tile's grid has no such pixel
  int orel[WM];    // the same for an OUTPUT pixel (not on the grid's left / right edge); far negative otherwise
  int ljt[WM];     // its column in the grid (the fold limit of the last tile column)
#pragma unroll
  for (int a = 0; a < WM; ++a) {
    const int m = (wm * WM + a) * 32 + l31;
    const int li = (int)(((unsigned)m * inv_w1) >> 20), lj = m - li * W1;
    arow1[a] = li < TH ? li * PW + lj : 0;
    hrel[a] = li < TH ? li * rowstride + lj : -(1 << 29);
    if constexpr (PAIR) orel[a] = (m >= 2 + d2 && m <= MT - 3 - d2) ? m : -(1 << 29);   // (pairs: index m = position base_h + m)
    else orel[a] = (li < TH && lj >= 1 && lj <= W1 - 2) ? li * rowstride + lj : -(1 << 29);
    ljt[a] = lj;
  }
  // the patch request of this thread: byte offset of its piece of patch row pr relative to the patch origin (kOob: no such row)
  unsigned voff_tab[NCQ + NHQ];
  int prow_h[NHQ];  // patch rows of the halo loads (the centre loads: lr8 + RQ q + c0)
#pragma unroll
  for (int q = 0; q < NCQ + NHQ; ++q) {
    int pr;
    if (q < NCQ) {
      pr = lr8 + RQ * q + c0;
    } else {
      const int hr = lr8 + RQ * (q - NCQ);
      pr = hr < c0 ? hr : hr + MT;
      prow_h[q - NCQ] = pr;
    }
    const int pi = (int)(((unsigned)pr * inv_pw) >> 20);
    const int rel = pi * rowstride + (pr - pi * PW);
    voff_tab[q] = pr < P ? (unsigned)rel * (unsigned)(C * 2) + 16u * c8 : kOob;
  }
  // LDS addresses of the patch writes: the 16-byte piece c8 of row r sits at slot c8 ^ ((r >> 1) & 7); rows RQ apart share a key
  const int wr_patch = R0 + (lr8 + c0) * ROWB + ((c8 ^ (((lr8 + c0) >> 1) & 7)) << 4);   // centre load q: + RQ q rows
  const int wr_raw = RX + lr8 * ROWB + ((c8 ^ ((lr8 >> 1) & 7)) << 4);                  // its raw copy at h pixel lr8 + RQ q
  int wr_halo[NHQ];
#pragma unroll
  for (int q = 0; q < NHQ; ++q) wr_halo[q] = R0 + prow_h[q] * ROWB + ((c8 ^ ((prow_h[q] >> 1) & 7)) << 4);

  // ---- tile cursor: (img, ti, tj) of the current tile and of the next one, advanced by increments -------------------------------------
  const int t_begin = blockIdx.x * per_block, t_end = min(t_begin + per_block, ntiles);
  // Folded layers walk DOWN the columns of tiles (tile row fastest): a 4 x 63 h tile reads 6 rows of d samples, two of them the rows the
  // tile above it read one tile earlier -- walked along the rows (35-odd tiles between vertical neighbours, ~50 MB through a 4 MB L2) every
  // tile fetched all six from HBM: 1.27-1.37 GB read per layer for a 0.91 GB tensor (PMC, profiles/r06_pmc.txt).
  constexpr bool down = DOWN;  // (a template argument: a run-time flag is three registers the 256-register singles do not have)
  int img = 0, ti = 0, tj = 0;
  if (t_begin < t_end) {
    if (down) {
      ti = t_begin % tiles_h;
      const int r = t_begin / tiles_h;
      tj = r % tiles_w;
      img = r / tiles_w;
    } else {
      tj = t_begin % tiles_w;
      const int r = t_begin / tiles_w;
      ti = r % tiles_h;
      img = r / tiles_h;
    }
  }
  auto advance = [&](int& im, int& i, int& j) __attribute__((always_inline)) {
    if (down) {
      if (++i == tiles_h) {
        i = 0;
        if (++j == tiles_w) {
          j = 0;
          ++im;
        }
      }
    } else if (++j == tiles_w) {
      j = 0;
      if (++i == tiles_h) {
        i = 0;
        ++im;
      }
    }
  };
  auto base_of = [&](int i, int j) __attribute__((always_inline)) {  // position of h pixel 0 (pairs: of index 0)
    const int j0 = j * TWo;
    return PAIR ? j0 - 2 - d2 : (p.fold ? i * TH * d + j0 - 1 : j0 - 1);
  };
  auto clip_len = [&](int im) __attribute__((always_inline)) {  // a varlen batch: the end of clip `im` (scalar load)
    return p.lens ? min(T, ((const VFX_CONST int*)p.lens)[im] * p.lens_mul) : T;
  };

  u32x4 PC[NCQ], PH[NHQ];  // the raw patch of the NEXT tile, in flight / landed
  auto request = [&](int im, int i, int j) __attribute__((always_inline)) {
    const int Tn = clip_len(im);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.x) + (int64_t)im * T * (C * 2)), 0, Tn * (C * 2), 0x00020000);
    const unsigned sbase = (unsigned)(base_of(i, j) - d) * (unsigned)(C * 2);  // patch origin (may be negative: wraps out of range)
#pragma unroll
    for (int q = 0; q < NCQ; ++q) PC[q] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(voff_tab[q] + sbase), 0, 0);
#pragma unroll
    for (int q = 0; q < NHQ; ++q) PH[q] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(voff_tab[NCQ + q] + sbase), 0, 0);
  };
  // 8 raw fp16 values -> LeakyReLU on the packed halves (max(x, slope x), 0 < slope <= 1)
  auto lrelu8 = [&](const u32x4& r) __attribute__((always_inline)) {
    const u32x2 a = f16x4_lrelu(u32x2{r[0], r[1]}, slope), b = f16x4_lrelu(u32x2{r[2], r[3]}, slope);
    return u32x4{a.x, a.y, b.x, b.y};
  };

  int Tb = T;
  VFX_TS_DECL;
  f32x16 y1r[WM];
  f32x16 acc[WM];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  auto mma = [&](const f16x8 (&w)[2], const char* img_base, const int (&row)[WM], int c) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f16x8 ah[WM];
#pragma unroll
      for (int a = 0; a < WM; ++a)
        ah[a] = *reinterpret_cast<const f16x8*>(img_base + row[a] * ROWB + ((64 * c + 32 * s + 16 * lh) ^ swz_key(row[a])));
#pragma unroll
      for (int a = 0; a < WM; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[s], ah[a], acc[a], 0, 0, 0);
    }
  };
  auto mma_set = [&](auto set_tag, int c, int k, const char* img_base, const int (&row)[WM]) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_tag)::value;
    if constexpr (PAIR && SET >= 1) {
      f16x8 w[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
        w[s2] = *reinterpret_cast<const f16x8*>(lds + WL_OFF + (((((SET - 1) * 6 + 3 * c + k) * 2 + s2) * 2 + wn) << 10) + lane * 16);
      mma(w, img_base, row, c);
    } else {
      mma(W[SET][c][k], img_base, row, c);
    }
  };

  // conv1 -> h -> conv2 of ONE layer over the index space of the tile (cf. k_resblock_rw); the accumulators hold conv2 on return
  auto layer = [&](auto second_tag, int base_h) __attribute__((always_inline)) {
    constexpr bool SECOND = decltype(second_tag)::value;
    const char* patch = lds + (SECOND ? R1 : R0);
    char* hbuf = lds + (SECOND ? R0 : R1);
    const float* bias1 = b1s + (SECOND ? 2 * C : 0);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        int rows[WM];
#pragma unroll
        for (int a = 0; a < WM; ++a) {
          if constexpr (SECOND) {
            const int r = (wm * WM + a) * 32 + l31 + (k - 1) * d2;
            rows[a] = r < 0 ? 0 : (r > MT - 1 ? MT - 1 : r);
          } else {
            rows[a] = arow1[a] + p.poff[k];
          }
        }
        mma_set(std::integral_constant<int, SECOND ? 2 : 0>{}, c, k, patch, rows);
      }
    VFX_TS(SECOND ? 9 : 4);  // conv1 done
    {
      unsigned sat16 = 0;
      const f16x2 slope2 = {(_Float16)slope, (_Float16)slope};
      u32x2 rres[WM][4];
      if constexpr (!SECOND) {
#pragma unroll
        for (int a = 0; a < WM; ++a) {
          const int m = (wm * WM + a) * 32 + l31;
          const char* rowx = lds + RX + m * ROWB + 8 * lh;
          const int key = (m >> 1) & 7;
#pragma unroll
          for (int j = 0; j < 4; ++j) rres[a][j] = *reinterpret_cast<const u32x2*>(rowx + (((wn * 4 + j) ^ key) << 4));
        }
        if constexpr (PAIR) __syncthreads();  // RX = R1 = the h buffer of this layer: every wave has its residual before any h is written
      }
#pragma unroll
      for (int a = 0; a < WM; ++a) {
        const int m = (wm * WM + a) * 32 + l31;
        const bool hval = (unsigned)(base_h + (SECOND ? m : hrel[a])) < (unsigned)Tb;
        char* rowp = hbuf + m * ROWB + 8 * lh;
        const int key = (m >> 1) & 7;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 b1v = *reinterpret_cast<const f32x4*>(bias1 + wn * 32 + 8 * j + 4 * lh);
          const unsigned h01 = lrelu_f16x2(pack_f16x2_sat16(acc[a][4 * j] + b1v[0], acc[a][4 * j + 1] + b1v[1], hval, sat16), slope2);
          const unsigned h23 = lrelu_f16x2(pack_f16x2_sat16(acc[a][4 * j + 2] + b1v[2], acc[a][4 * j + 3] + b1v[3], hval, sat16), slope2);
          if constexpr (!SECOND) {  // conv2 accumulates on top of the residual
            const f32x4 v = f16x4_widen(rres[a][j]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[a][4 * j + e] = v[e];
          } else {                  // a pair's second layer: on top of y1
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[a][4 * j + e] = y1r[a][4 * j + e];
          }
          *reinterpret_cast<uint2*>(rowp + (((wn * 4 + j) ^ key) << 4)) = make_uint2(h01, h23);
        }
      }
      report_f16_saturation(f16_sat16_bad(sat16), p.flags);
    }
    VFX_TS(SECOND ? 10 : 5);  // residual read, h written
    __syncthreads();  // h is complete
    VFX_TS(SECOND ? 11 : 6);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        int rows[WM];
#pragma unroll
        for (int a = 0; a < WM; ++a) {
          const int r = (wm * WM + a) * 32 + l31 + k - 1;
          rows[a] = r < 0 ? 0 : (r > MT - 1 ? MT - 1 : r);  // clamped rows only feed outputs that are masked anyway
        }
        mma_set(std::integral_constant<int, SECOND ? 3 : 1>{}, c, k, hbuf, rows);
      }
    if constexpr (PAIR) __syncthreads();  // (single layers: see k_resblock_rw -- the next patch write is ordered by the next tile's barriers)
  };

  if (t_begin < t_end) request(img, ti, tj);
  __syncthreads();  // the biases (and the fragments a pair keeps in LDS) are there
  for (int t = t_begin; t < t_end; ++t) {
    VFX_TS(0);
    const int base_h = base_of(ti, tj);
    const int j0 = tj * TWo;
    int nimg = img, nti = ti, ntj = tj;  // the next tile
    advance(nimg, nti, ntj);
    if (p.lens) {
      Tb = clip_len(img);
      if ((PAIR ? j0 : base_h + 1) >= Tb) {  // the tile lies wholly past the end of its clip: nothing to compute or store
        if (t + 1 < t_end) request(nimg, nti, ntj);
        img = nimg, ti = nti, tj = ntj;
        continue;
      }
    }
    // ---- the landed patch: to LDS as operands, its centre rows also raw (the residual) -------------------------------------------------
#pragma unroll
    for (int q = 0; q < NCQ; ++q) {
      *reinterpret_cast<u32x4*>(lds + wr_raw + q * RQ * ROWB) = PC[q];
      *reinterpret_cast<u32x4*>(lds + wr_patch + q * RQ * ROWB) = lrelu8(PC[q]);
    }
#pragma unroll
    for (int q = 0; q < NHQ; ++q) *reinterpret_cast<u32x4*>(lds + wr_halo[q]) = lrelu8(PH[q]);
    VFX_TS(1);  // patch rows written
    __syncthreads();  // the patch is complete
    VFX_TS(2);
    if (t + 1 < t_end) request(nimg, nti, ntj);  // lands while this tile is computed and stored
    VFX_TS(3);  // next patch requested

    layer(std::false_type{}, base_h);
    VFX_TS(7);  // first (only) layer: conv2 done

    if constexpr (PAIR) {
      // ---- between the layers: y1 = acc (conv2 on top of x) + b2 stays in registers; LeakyReLU(y1) becomes the second layer's operand
      // rows in R1 (index m at row m; zero outside the sequence and on the two indices where y1 is not valid) ---------------------------
      unsigned sat16 = 0;
#pragma unroll
      for (int a = 0; a < WM; ++a) {
        const int m = (wm * WM + a) * 32 + l31;
        const bool ok = (m >= 1) & (m <= MT - 2) & ((unsigned)(base_h + m) < (unsigned)Tb);
        char* rowp = lds + R1 + m * ROWB + 8 * lh;
        const int key = (m >> 1) & 7;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 b2a = *reinterpret_cast<const f32x4*>(b1s + C + wn * 32 + 8 * j + 4 * lh);
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float tt = acc[a][4 * j + e] + b2a[e];
            y1r[a][4 * j + e] = tt;
            acc[a][4 * j + e] = 0.f;
            v[e] = fmaxf(tt, tt * slope);
          }
          *reinterpret_cast<uint2*>(rowp + (((wn * 4 + j) ^ key) << 4)) =
              make_uint2(pack_f16x2_sat16(v[0], v[1], ok, sat16), pack_f16x2_sat16(v[2], v[3], ok, sat16));
        }
      }
      report_f16_saturation(f16_sat16_bad(sat16), p.flags);
      __syncthreads();  // the second patch is complete
      VFX_TS(8);
      layer(std::true_type{}, base_h);
      VFX_TS(12);  // second layer: conv2 done
    }

    // ---- epilogue: y = acc + b2 (the residual is inside the accumulators) straight to memory, through descriptors of the clip ------------
    {
      const int ybytes = Tb * (C * 2);
      const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
          p.y ? reinterpret_cast<char*>(p.y) + (int64_t)img * T * (C * 2) : nullptr, 0, p.y ? ybytes : 0, 0x00020000);
      const __amdgpu_buffer_rsrc_t rya = __builtin_amdgcn_make_buffer_rsrc(
          p.ya ? reinterpret_cast<char*>(p.ya) + (int64_t)img * T * (C * 2) : nullptr, 0, p.ya ? ybytes : 0, 0x00020000);
      const bool have_y = p.y != nullptr, have_ya = p.ya != nullptr;
      const float aslope = p.act_slope;
      unsigned sat16 = 0;
      f32x4 b2r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b2r[j] = *reinterpret_cast<const f32x4*>(b1s + (PAIR ? 3 * C : C) + wn * 32 + 8 * j + 4 * lh);
      const unsigned lane_off = (unsigned)(wn * 64 + 16 * lh);
#pragma unroll
      for (int a = 0; a < WM; ++a) {
        // an output pixel of the tile's grid, inside the row of d samples (folded tiles: the last tile column may stick out)
        const bool ok = (orel[a] >= 0) & (PAIR | !p.fold | (j0 + ljt[a] - 1 < d));
        const unsigned rowoff = ok ? (unsigned)(base_h + orel[a]) * (unsigned)(C * 2) + lane_off : kOob;  // (past the clip: out of range)
#pragma unroll
        for (int jp = 0; jp < 4; jp += 2) {
          f32x4 v[2];
#pragma unroll
          for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[r][e] = acc[a][4 * (jp + r) + e] + b2r[jp + r][e];
              acc[a][4 * (jp + r) + e] = 0.f;
            }
          // lanes 0-31 keep their run jp and receive the partner's run jp; lanes 32-63 receive the partner's run jp + 1 and keep theirs
          if (have_y) {
            const auto s0 = __builtin_amdgcn_permlane32_swap(pack_f16x2_sat16(v[0][0], v[0][1], true, sat16), pack_f16x2_sat16(v[1][0], v[1][1], true, sat16), false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(pack_f16x2_sat16(v[0][2], v[0][3], true, sat16), pack_f16x2_sat16(v[1][2], v[1][3], true, sat16), false, false);
            const u32x4 w = {s0[0], s1[0], s0[1], s1[1]};
#ifdef VFX_ABL_NOSTORE  // timing-only build (wrong results): what the direct stores cost
            asm volatile("" : : "v"(w));
#else
            __builtin_amdgcn_raw_buffer_store_b128(w, ry, (int)(rowoff + (unsigned)(16 * jp)), 0, 0);
#endif
          }
          if (have_ya) {  // last layer in front of an upsampler: ya = fp16(LeakyReLU(y, act_slope))
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
      report_f16_saturation(f16_sat16_bad(sat16), p.flags);
    }
    VFX_TS(13);  // stores issued
    VFX_TS_FLUSH(p.timing, t, wave_u, NW);
    img = nimg, ti = nti, tj = ntj;
  }
}

// h positions per tile of the register-weights kernel: 256; 0 = off (VFX_TUNE_NO_PERSISTENT_C64: k_resblock runs the layer)
int resblock_rw_tile(int tuning) { return (tuning & VFX_TUNE_NO_PERSISTENT_C64) ? 0 : 256; }

// Two consecutive layers as one launch: 256-position tiles, the first layer's patch must fit (d <= 32) and a tile must still
// advance by at least half of its positions (d2 <= 62).  VFX_TUNE_NO_PAIRS: one launch per layer.
bool resblock_rw_pair_ok(int C, int dil, int dil2, int tuning) {
  return !(tuning & VFX_TUNE_NO_PAIRS) && C == 64 && resblock_rw_tile(tuning) == 256 && dil >= 1 && dil <= 32 && dil2 >= 1 &&
         256 - 4 - 2 * dil2 >= 128;
}

int cu_count_of_current_device() {
  static int cus[64] = {};
  int dev = 0;
  VFX_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) dev = 0;
  if (cus[dev] == 0) {
    int n = 0;
    VFX_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    cus[dev] = n > 0 ? n : 256;
  }
  return cus[dev];
}

template <bool PAIR, int HALO = 64, bool DOWN = false>
static void launch_rw16(const ResBlockParams* dparams, int64_t ntiles, hipStream_t stream) {  // DOWN: see the tile cursor
  constexpr int MT = 256;
  // the layout of k_resblock_rw<8, PAIR, true, HALO>: the two operand regions, the raw rows (singles: a third region), the biases,
  // a pair's three sets of weight fragments
  const size_t lds = (size_t)(MT + HALO + MT) * 128 + (PAIR ? (size_t)72 * 1024 + 4 * 64 * sizeof(float) : (size_t)MT * 128 + 2 * 64 * sizeof(float));
  const int slots = cu_count_of_current_device();  // 256 registers per wave: one 8-wave block per CU
  const int per_block = (int)((ntiles + slots - 1) / slots);
  const int grid = (int)((ntiles + per_block - 1) / per_block);
  static uint64_t attr_devices = 0;
  if (first_use_on_current_device(attr_devices)) {
    VFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_resblock_rw16<PAIR, HALO, DOWN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  hipLaunchKernelGGL((k_resblock_rw16<PAIR, HALO, DOWN>), dim3(grid), dim3(512), lds, stream, dparams, (int)ntiles, per_block);
}

template <int NW, bool PAIR, bool X16, int HALO = 64>
static void launch_rw(const ResBlockParams* dparams, int64_t ntiles, hipStream_t stream) {
  constexpr int MT = NW * 32;
  // the two operand regions (the staged accumulators overlay them) + biases; pairs: + three sets of weight fragments
  // (DIRECT = single layers on the fp16 trunk: + the raw rows RX and b2)
  const size_t lds = (size_t)(MT + HALO + MT) * 128 + (PAIR ? (size_t)72 * 1024 + 4 * 64 * sizeof(float)
                                                             : (X16 ? (size_t)MT * 128 + 2 * 64 * sizeof(float) : 64 * sizeof(float)));
  const int slots = cu_count_of_current_device() * (NW == 4 ? 2 : 1);  // 256 registers per wave: 8 waves per CU
  const int per_block = (int)((ntiles + slots - 1) / slots);
  const int grid = (int)((ntiles + per_block - 1) / per_block);
  static uint64_t attr_devices = 0;
  if (first_use_on_current_device(attr_devices)) {
    VFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_resblock_rw<NW, PAIR, X16, HALO>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  hipLaunchKernelGGL((k_resblock_rw<NW, PAIR, X16, HALO>), dim3(grid), dim3(NW * 64), lds, stream, dparams, (int)ntiles, per_block);
}

// measurement builds (-DVFX_RW_OLD16): the fp16-trunk layers on k_resblock_rw<.., X16> as in round 5 (same-box A/B of k_resblock_rw16)
#ifdef VFX_RW_OLD16
constexpr bool kOldRw16 = true;
#else
constexpr bool kOldRw16 = false;
#endif

void launch_resblock_rw(const ResBlockParams& hp, const ResBlockParams* dparams, hipStream_t stream) {
  VFX_CHECK(hp.rw && hp.hionly && hp.C == 64 && !hp.geo2d && !hp.asrc, "resblock_rw: needs the 16-bit mode and C = 64");
  VFX_CHECK(!hp.lens || hp.x16, "resblock_rw: a batch of clips of unequal length needs the fp16 trunk of the 16-bit mode (not with "
            "VFX_TUNE_F32_TRUNK, and 0 < voc_res_slope <= 1); use precision 1 otherwise");
  const int64_t ntiles = (int64_t)hp.B * hp.tiles_h * hp.tiles_w;
  VFX_CHECK(ntiles > 0 && ntiles < ((int64_t)1 << 30), "resblock_rw: bad tile count");
  VFX_CHECK(hp.x && (hp.y || (hp.x16 && hp.ya)), "resblock_rw: no input / no output");
  VFX_CHECK(!hp.x16 || (hp.slope > 0.f && hp.slope <= 1.f), "resblock_rw: the packed LeakyReLU of the fp16 trunk needs 0 < slope <= 1");
  if (hp.dil2 > 0) {
    VFX_CHECK(hp.tile_m == 256 && !hp.fold && hp.w1b && hp.w2b && hp.b1b && hp.b2b, "resblock_rw: bad layer pair");
    if (hp.x16 && !kOldRw16) launch_rw16<true>(dparams, ntiles, stream);
    else if (hp.x16) launch_rw<8, true, true>(dparams, ntiles, stream);
    else launch_rw<8, true, false>(dparams, ntiles, stream);
  } else if (hp.tile_m == 256) {
    VFX_CHECK(hp.patch_rows == 0 || (hp.patch_rows == 256 + 128 && hp.x16 && hp.fold), "resblock_rw: bad patch geometry");
#ifdef VFX_RW16_ROWMAJOR  // measurement builds: folded tiles walked along the rows, as before
    if (hp.x16 && hp.patch_rows && !kOldRw16) launch_rw16<false, 128>(dparams, ntiles, stream);
#else
    if (hp.x16 && hp.patch_rows && !kOldRw16) launch_rw16<false, 128, true>(dparams, ntiles, stream);  // (the wide tiles are folded ones)
    else if (hp.x16 && hp.fold && !kOldRw16) launch_rw16<false, 64, true>(dparams, ntiles, stream);
#endif
    else if (hp.x16 && !kOldRw16) launch_rw16<false>(dparams, ntiles, stream);
    else if (hp.x16 && hp.patch_rows) launch_rw<8, false, true, 128>(dparams, ntiles, stream);
    else if (hp.x16) launch_rw<8, false, true>(dparams, ntiles, stream);
    else launch_rw<8, false, false>(dparams, ntiles, stream);
  } else if (hp.tile_m == 128 && hp.x16 && !hp.patch_rows) {
    launch_rw<4, false, true>(dparams, ntiles, stream);  // two 4-wave blocks per CU (measurement builds: -DVFX_RW_SINGLE_MT=128)
  } else VFX_CHECK(false, "resblock_rw: tile of %d positions", hp.tile_m);
  VFX_HIP(hipGetLastError());
}

}  // namespace vfx
