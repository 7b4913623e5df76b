// conv.hip -- stage-driven tap convolution (implicit GEMM) for gfx950.
//
// One kernel serves every GEMM-shaped layer of the hot path:
//   * 3x3 Conv2d of ConvBlockRes (models/components/modules.py:223-271) incl. its BN+LeakyReLU
//     pre-activation (prologue), the identity / 1x1-conv shortcut (extra K segment on the raw
//     input) and the residual add (epilogue);
//   * the channel concat in DecoderBlockRes4B (modules.py:212-220) as two K segments;
//   * the stride-2 ConvTranspose2d of the decoders as 4 output-parity launches;
//   * the TFGAN vocoder's Conv1d k3 (dilated; folded into rows of d samples when d is wide) / k7 (reflect
//     padded) and its ConvTranspose1d upsamplers as ONE phased launch each (output phases = cout ranges with
//     their own stage tables, TapConvParams::nphase).
//
// GEMM view:  out[m, n] = sum_seg sum_tap sum_c  P(src_seg[pix(m) + off(tap), c]) * W_seg[tap][c][n]
//   M = output pixels, N = Cout, K = sum ntaps*C.  Activations are channels-last, 4 bytes per
//   element, so a K step (one tap, 32 channels) is a 128-byte row per pixel.
//
// Work decomposition
//   block   = 4 waves, tile = TH x TW (<= 128) pixels of ONE image x BN in {128, 64, 32} couts;
//   wave    = (128 / WAVES_M) pixels x 32 couts, WAVES_N = BN / 32;
//   (round 4 measured a BN = 256 tile with 64-cout waves for the 16-bit launches on activated sources -- half the LDS fragment
//   reads per MFMA, two blocks per CU instead of three: +-0 on the C = 512 stack and the upsamplers, bit-identical results;
//   deleted, numbers in profiles/r04_c5_wide_conv_tile_ab.jsonl)
//   stage   = one 32-channel chunk of one source tensor and the taps that read it.  The host
//             flattens every launch into a table of stages (ConvStage, vfx_internal.h).
//
// Operand paths
//   A (activations): per stage the input PATCH (tile + halo of the stage's taps, <= 192 pixels x
//     128 bytes) goes into one of two LDS patch buffers; every tap reads its shifted window of it.
//     - ACTIVATED sources (the producer's epilogue already applied this layer's prologue and
//       stored the MFMA operand form, TapConvParams::out_act) are copied global -> LDS by the
//       LDS-DMA engine (buffer_load ... lds): no VGPRs, no VALU, no ds_write; pixels outside the
//       image read past the buffer bound and arrive as zeros.
//     - RAW fp32 sources arrive the same way and are then transformed IN PLACE by the threads that
//       fetched them: per-channel affine (folded eval-mode BatchNorm), LeakyReLU / ReLU / ELU, zero
//       halo AFTER the activation, hi/lo split.
//     The patch of stage s+1 is requested while stage s computes; ONE barrier per stage.
//     LDS rows are unpadded 128-byte rows (the DMA writes lane-linear); the 16-byte piece p of row r
//     sits at slot p ^ ((r >> 1) & 7), applied on the DMA's SOURCE address and on every read, which
//     makes the 16-lane groups of the ds_read_b128 fragment reads conflict-free.
//   B (weights): never touch LDS.  Packed on the host in MFMA fragment order, they go
//     global -> VGPR -> MFMA through a ring of three (two for BN = 32) register groups.  The loads
//     are inline asm with hand-counted s_waitcnt vmcnt(N): beside an LDS-DMA the compiler would
//     wait vmcnt(0) for every ordinary load and drain the patch prefetch at each tap.
//     Until its wait a ring register holds stale data and the compiler does not know.  Rules kept here and
//     checked on the assembly by scripts/asm_inflight_check.py (tests/test_host.py):
//       - conditional waits carry no register operands (a merge copy could be placed before the wait);
//       - a group becomes readable at ONE unconditional use_b() between its wait and its first MFMA;
//       - a group is never copied (R1 = R0) while its load is in flight: the copy reads stale registers and
//         makes both groups the same value for the compiler, which may then feed the stale one to the MFMAs.
//
// Arithmetic (vfx_config.precision; 2 = as 1, except that the vocoder's launches set TapConvParams::hionly: the hi
// halves of weights and activations hold fp16 values and are the only ones loaded and multiplied -- one
// v_mfma_f32_32x32x16_f16 per product, fp32 accumulate)
//   1: split-bf16 -- every operand is hi + lo (two bf16), products hi*hi + hi*lo + lo*hi on
//      v_mfma_f32_32x32x16_bf16, fp32 accumulate (~2^-16 relative operand error);
//   0: exact fp32 on v_mfma_f32_32x32x2_f32.
//   Plain 16-bit operands miss the "log-mel L1 <= 1e-3" bar in the ResUNets (bf16 19x, fp16 3.3x); the vocoder
//   holds 58 dB SI-SDR on fp16 and only 40 dB on bf16.
#include "conv_common.h"
#include "conv_epilogue.h"
#include "vfx_internal.h"

namespace vfx {

// ABL != 0: timing-only ablation builds (-DVFX_ABLATION_BUILD + VFX_ABLATE, wrong results), in the stage loop:
// bit 0 no patch request, 1 no weight loads, 2 no barrier, 3 constant fragment addresses, 4 no fragment reads,
// 5 no epilogue, 6 the end-of-stage wait leaves the patch in flight.
// HI: 16-bit (fp16) operands, one MFMA per product (TapConvParams::hionly).  H64 (with HI): every source of the launch is
// an ACTIVATED fp16 tensor with 2-byte elements -- a 128-byte patch row holds 64 channels, a stage is a 64-channel
// chunk, a tap is four K = 16 steps on four fp16 weight fragments (half the stages, barriers and DMA instructions of
// the 32-channel form, and every fetched byte is an operand).  HI without H64: raw fp32 sources, 32-channel stages,
// transformed in place to fp16 in the first half of the row, two K = 16 steps per tap.
// RA (with H64): the launch's residual is an activated fp16 tensor, inverted in the epilogue (TapConvParams::residual_act).
// Timing builds (-DVFX_TIMING, scripts/conv_timing.py): per wave, shader-clock stamps at entry / setup done / first patch + weights
// landed / first transform done / tap loop done / epilogue done, and the chip-wide 100 MHz clock at entry and exit; inside the tap loop
// the cycles summed over the taps of [barrier + fetch + patch request + weight wait], [fragment reads + MFMAs], [end of stage: wait,
// transform; cursor updates]: timing[(block * 4 + wave) * 12 + 0 .. 10]
#ifdef VFX_TIMING
#define KCONV_TIMING_PARAM , unsigned long long* __restrict__ kc_timing
#define KCONV_TS_BEGIN() unsigned long long kc_ts[6] = {}, kc_acc[3] = {}, kc_a = 0, kc_b = 0; const unsigned long long kc_rt0 = __builtin_amdgcn_s_memrealtime(); kc_ts[0] = __builtin_readcyclecounter()
#define KCONV_TAP_A() kc_a = __builtin_readcyclecounter()
#define KCONV_TAP(i) do { kc_b = __builtin_readcyclecounter(); kc_acc[i] += kc_b - kc_a; kc_a = kc_b; } while (0)
#define KCONV_TS(i) kc_ts[i] = __builtin_readcyclecounter()
#define KCONV_TS_END()                                                                                           \
  do {                                                                                                           \
    if (kc_timing && (threadIdx.x & 63) == 0) {                                                                  \
      unsigned long long* tp_ = kc_timing + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 12;                 \
      for (int i_ = 0; i_ < 6; ++i_) tp_[i_] = kc_ts[i_];                                                        \
      tp_[6] = kc_rt0;                                                                                           \
      tp_[7] = __builtin_amdgcn_s_memrealtime();                                                                 \
      for (int i_ = 0; i_ < 3; ++i_) tp_[8 + i_] = kc_acc[i_];                                                   \
    }                                                                                                            \
  } while (0)
#else
#define KCONV_TIMING_PARAM
#define KCONV_TS_BEGIN()
#define KCONV_TS(i)
#define KCONV_TS_END()
#define KCONV_TAP_A()
#define KCONV_TAP(i)
#endif
template <int BN, bool ELU, bool SPLIT, int ABL = 0, int RING = 3, bool HI = false, bool H64 = false, bool RA = false, bool VL = false>
__global__ __launch_bounds__(256, (BN <= 64 || HI) ? 3 : 2) void k_conv(const TapConvParams* __restrict__ pp KCONV_TIMING_PARAM) {
  KCONV_TS_BEGIN();
  static_assert(!H64 || (HI && SPLIT), "H64 is a variant of the 16-bit mode");
  static_assert(!RA || H64, "an activated residual exists in the 16-bit mode's activated-source launches only");
  constexpr bool HI32 = HI && !H64;  // the hi fragments (f[0], f[2]) only
  constexpr int WNB = 1;                      // 32-cout blocks per wave
  constexpr int WAVES_N = BN / 32;
  constexpr int WM = BN / 32;                 // 32-row blocks per wave (= 4 / WAVES_M)

  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int EPI_HALVES = (BN == 128 && HI) ? 2 : 1;  // 16-bit BN = 128 tile: half-width epilogue staging -> three blocks per CU
  constexpr int kEpiBytes = CBM * (BN / EPI_HALVES + 4) * 4;
  constexpr int kMainBytes = (2 * CPATCH > kEpiBytes) ? 2 * CPATCH : kEpiBytes;
  char* const lds = reinterpret_cast<char*>(smem);
  int* otab = reinterpret_cast<int*>(lds + kMainBytes);  // [kOtabSlots][128] output pixel index or -1 (one slot per phase of the block)

  const TapConvParams& p = *pp;
  // The stage table is read-only for the whole launch: address it in the constant address space so
  // that every descriptor field is a scalar load.
  typedef const ConvStage VFX_CONST* StageTab;
  const int tid = threadIdx.x;
  const int n_tiles = p.Cout / BN;
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch rule, speed only), so
  // give every XCD a contiguous range of tiles -- neighbouring tiles share halo rows and the
  // N-tiles of one spatial tile share the whole patch through that XCD's L2.
  int tile;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  // split-K: block (tile, ks) walks stage range ks of the launch (TapConvParams::ksplit)
  const int KS = p.ksplit > 1 ? p.ksplit : 1;
  const int ks = tile % KS;
  tile /= KS;
  const int n0 = (tile % n_tiles) * BN;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave_u / WAVES_N, wn = wave_u % WAVES_N;
  // phased launch: the 32 couts of a WAVE belong to phase (n0 + 32 wn) / cout_phase -- own stage table (own taps), own weight
  // tensor.  The waves of a block may belong to different phases (BN a multiple of cout_phase, round 6: the ResUNets' 2 x upsamplers,
  // whose phases are 32-64 couts wide -- one block stages the patch once for all of them); the patch, its prologue and the stage
  // count are the same in every phase, the only block-wide synchronisation is the barrier at the start of a stage, so waves with
  // fewer taps simply arrive early.
  const int phase0 = p.nphase > 1 ? n0 / p.cout_phase : 0;              // first phase of the block
  const int phase = p.nphase > 1 ? (n0 + 32 * wn) / p.cout_phase : 0;   // this wave's
  const int n0w = p.nphase > 1 ? n0 + 32 * wn - phase * p.cout_phase : n0 + 32 * wn;  // the wave's first cout inside its weight tensor
  const int st_lo = (int)((int64_t)ks * p.nstages / KS), st_hi = (int)((int64_t)(ks + 1) * p.nstages / KS);
  const StageTab stages = (StageTab)(uintptr_t)p.stages + phase * p.nstages + st_lo;
  int mt = tile / n_tiles;  // spatial tile: (image, tile row, tile col), col fastest
  const int tj = mt % p.tiles_w;
  mt /= p.tiles_w;
  const int ti = mt % p.tiles_h;
  const int img = mt / p.tiles_h;
  const int i0 = ti * p.TH, j0 = tj * p.TW;
  const int Hi = p.Hi, PW = p.PW, P = p.P;
  const int in_img_stride = p.in_img_stride;
  // VL: batches of clips of unequal length (TapConvParams::lens; the vocoder's 1-D launches): clip `img` ends at its own
  // length -- positions past it read as zeros (or reflect at the clip's own end) and are not written, exactly what a
  // batch-of-one call of that clip computes; a tile that lies wholly past the end has nothing to do.  A variant of its own:
  // the per-clip limits are computed values that stay live through the stage loop, where the uniform ones are re-read from
  // the parameter block -- three registers the 168-register tiles of the ResUNet launches do not have.
  int in_limit = p.in_limit, out_limit = p.out_limit, Wi = p.Wi;
  if constexpr (VL) {
    const int n = ((const VFX_GLOBAL int*)p.lens)[__builtin_amdgcn_readfirstlane(img)];
    in_limit = min(in_limit, n * p.lens_mul_in);
    out_limit = min(out_limit, n * p.lens_mul_out);
    if (Hi == 1) Wi = in_limit;
    if ((i0 * p.sh + p.oh0) * p.Wo + j0 * p.sw + p.ow0 >= out_limit) return;
  }
  const int nq = (P + 31) >> 5;  // patch row groups in use (uniform)

  // ---- per-thread roles ---------------------------------------------------------------------------
  // Every stage of a launch stages the same PH x PW window around the tile; only its origin (dh0, dw0)
  // varies between stages.
  const int lr = tid >> 3, cg = tid & 7;
  const int tw_shift = p.tw_shift, TWm1 = p.TW - 1, TH = p.TH;
  // Swizzle key of patch pixel (pi, pj): ((pj >> 1) + (TW / 2) * pi) & 7.  With an even patch width the bank half of
  // LDS row pi*PW + pj is pj & 1, and the 16 lanes of a ds_read_b128 group -- runs of consecutive columns in TW-wide
  // tile rows -- get 16 distinct (bank half, slot) pairs for every tap shift; for a 1-D patch this is (row >> 1) & 7.
  // The 16-bit BN = 128 tile (the vocoder's 1-D layers only, three blocks per CU: short of registers) keeps the 1-D form
  // (LDS row >> 1) & 7, cheaper per tap.  The split / fp32 BN = 128 tile serves ResUNet levels 2 and 3 on 16 x 8 tiles,
  // where the 1-D key makes EVERY fragment read a 3-way conflict (scripts/lds_conflicts_conv.py; PMC: 64 % of the LDS
  // cycles) -- a wave of this tile reads all 128 pixel fragments per K step, 85 B/clk/CU at full MFMA rate before conflicts.
  constexpr bool SWZ2D = BN <= 64 || !HI;
  const int hTW = p.TW >> 1;
  // prow / PW as multiply-shift (prow < 192 = kPatchMaxRows, PW <= 192: exact): keyq[] and set_origin() were twelve integer
  // divisions of ~25 VALU instructions per thread and block (round 6)
  const unsigned inv_pw = ((1u << 20) + (unsigned)PW - 1u) / (unsigned)PW;
  int keyq[CNQ];  // key of this thread's patch pixels lr + 32q
#pragma unroll
  for (int q = 0; q < CNQ; ++q) {
    const int prow = lr + 32 * q;
    const int pi = (int)(((unsigned)prow * inv_pw) >> 20), pj = prow - pi * PW;
    keyq[q] = SWZ2D ? ((pj >> 1) + hTW * pi) & 7 : (prow >> 1) & 7;
  }
  if (tid < CBM) {
    const int li = tid >> tw_shift, lj = tid & TWm1;
    const int i = i0 + li, j = j0 + lj;
    // one table per phase of the block (conv_epilogue picks the slot of its couts); ordinary launches: one
    const int nslots = (p.nphase > 1 && p.cout_phase < BN) ? BN / p.cout_phase : 1;
    for (int s = 0; s < nslots; ++s) {
      const int ph = phase0 + s;
      int idx = -1;
      if (li < TH && i < p.Hg && j < p.Wg) {
        // (out_cmul: phase r of an odd-width phased launch writes true column ow + r -- the epilogue's channel offset r * C of a
        // tensor addressed in units of C; phase_rows: phases 2 and 3 are the odd output rows, columns r & 1: the pixel index takes
        // the row and gives back the two columns the channel offset adds)
        const int prow = p.phase_rows ? ph >> 1 : 0, pcol = p.out_cmul ? (p.phase_rows ? ph & 1 : ph) : 0;
        const int oh = i * p.sh + p.oh0 + prow, ow = j * p.sw + p.ow0;
        const int o = oh * p.Wo + ow;
        if (oh < p.Ho && ow + pcol < p.Wo && o < out_limit) idx = img * p.out_img_stride + o - 2 * prow;
      }
      otab[s * CBM + tid] = idx;
    }
  }

  f32x16 acc[WM][WNB];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int b = 0; b < WNB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int lane = tid & 63;
  const int l31 = lane & 31, lh = lane >> 5;
  int arow[WM];  // patch row of this lane's pixel of M block a (tap offset added per step)
  int ak0[WM];   // its swizzle key before the tap shift: (lj >> 1) + (TW / 2) * li, column parity in bit 16
#pragma unroll
  for (int a = 0; a < WM; ++a) {
    const int ml = (wm * WM + a) * 32 + l31;
    const int li = ml >> tw_shift, lj = ml & TWm1;
    arow[a] = li < TH ? li * PW + lj : 0;
    ak0[a] = li < TH ? (((lj >> 1) + hTW * li) | ((lj & 1) << 16)) : 0;
  }
  const unsigned nb_off = (unsigned)((n0w >> 5) * 1024 + lane * 4) * 4u;  // byte offset of this lane in a fragment block

  // ---- patch (A) staging ------------------------------------------------------------------------
  // The byte offset of every patch pixel is kept in registers and only recomputed when the patch
  // origin or the pixel stride changes (per-tap and mixed-width launches only).
  unsigned voff[CNQ];  // pixel byte offset inside the source tensor (chunk 0), valid when the okmask bit is set
  unsigned okmask = 0;
  int o_dh = 0x7fffffff, o_dw = 0, o_C = 0;  // origin / stride the offsets were computed for
  bool praw = false;   // the patch in flight is a raw fp32 one: transformed in place once it has landed
  auto set_origin = [&](int dh, int dw, int C) __attribute__((always_inline)) {
    o_dh = dh;
    o_dw = dw;
    o_C = C;
    okmask = 0;
#pragma unroll
    for (int q = 0; q < CNQ; ++q) {
      const int prow = lr + 32 * q;  // patch pixel -> (row, col) of the PH x PW window
      const int pi = (int)(((unsigned)prow * inv_pw) >> 20), pj = prow - pi * PW;
      const int si = prow < P ? i0 + dh + pi : -1;
      int sj = j0 + dw + pj;
      int rj = sj < 0 ? -sj : sj;
      rj = rj >= Wi ? 2 * (Wi - 1) - rj : rj;
      sj = p.reflect_w ? rj : sj;
      const int o = si * Wi + sj;
      const bool ok = ((unsigned)si < (unsigned)Hi) & ((unsigned)sj < (unsigned)Wi) & (o < in_limit);
      voff[q] = (unsigned)(img * in_img_stride + o) * (unsigned)(C * 4);
      okmask |= ok ? (1u << q) : 0u;
    }
  };

  // Request the patch of stage S into the LDS buffer at byte offset `dst`: CNQ LDS-DMA instructions (the
  // hand-counted weight waits rely on that number).  Offsets of pixels outside the image / patch are
  // replaced by one past the buffer bound: the hardware returns zeros.  A thread's lane always lands in
  // slot cg of patch row lr + 32q; an activated source is fetched pre-swizzled (piece cg ^ key), a raw one
  // straight (piece cg) -- the same thread re-reads and rewrites it in transform_patch().
  auto issue_patch = [&](const ConvStage VFX_CONST& S, int dst) __attribute__((always_inline)) {
    const int dh = S.dh0, dw = S.dw0, C = S.C;
    if (dh != o_dh || dw != o_dw || C != o_C) set_origin(dh, dw, C);  // uniform; VALU only
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((const float*)S.src), 0, (int)S.nbytes, 0x00020000);
    praw = (S.flags & 1) == 0;
#pragma unroll
    for (int q = 0; q < CNQ; ++q) {
      // the lane's bytes always land in slot cg: an activated source is fetched pre-swizzled (piece cg ^ key)
      const unsigned piece = praw ? 16u * cg : (unsigned)(cg ^ keyq[q]) << 4;
      // 16-bit mode: only the hi half (pieces 0..3) of an activated row is ever read -- the other lanes fetch nothing
      const bool need = !HI32 || praw || piece < 64u;
      const unsigned o = ((okmask & (1u << q)) && need) ? voff[q] + piece : 0xfffffff0u;
      VFX_LDS void* l = (VFX_LDS void*)(lds + dst + (32 * q + 8 * wave_u) * CROW);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, (int)o, 0, 0, 0);
    }
  };

  // Raw source: every thread turns the 4 floats it fetched (slot cg of its rows) into operand form in
  // place: per-channel affine (folded eval-mode BatchNorm), LeakyReLU / ReLU / ELU, zero halo AFTER the
  // activation, hi/lo split.  All reads of a thread precede its writes and a row's 8 slots belong to 8
  // consecutive lanes of one wave, so no barrier is needed between the two passes.
  auto transform_patch = [&](const ConvStage VFX_CONST& S, int dst) __attribute__((always_inline)) {
    const f32x4 psc = *(const VFX_GLOBAL f32x4*)((const float*)S.scale + 4 * cg);
    const f32x4 psh = *(const VFX_GLOBAL f32x4*)((const float*)S.shift + 4 * cg);
    const float pslope = S.slope;
    char* row0 = lds + dst + lr * CROW;
    f32x4 raw[CNQ];
    unsigned f16_sat = 0;  // 16-bit mode: a value left the fp16 range and was clamped (reported per patch: no loop-carried state)
#pragma unroll
    for (int q = 0; q < CNQ; ++q)
      if (q < 4 || q < nq) raw[q] = *reinterpret_cast<const f32x4*>(row0 + 32 * q * CROW + 16 * cg);
#pragma unroll
    for (int q = 0; q < CNQ; ++q)
      if (q < 4 || q < nq) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = raw[q][e] * psc[e] + psh[e];
          float u;
          if constexpr (ELU) u = t > 0.f ? t : expm1f(t);  // only vfx_op_conv instantiates this
          else u = fmaxf(t, t * pslope);                    // LeakyReLU, slope in [0, 1]: 1 = identity, 0 = ReLU
          v[e] = (okmask & (1u << q)) ? u : 0.f;
        }
        char* rowp = row0 + 32 * q * CROW;
        if constexpr (SPLIT && HI) {
          // fp16 values of channels 4cg..4cg+3 in the hi half of the row: piece cg>>1, half cg&1 (lo pieces unused)
          *reinterpret_cast<uint2*>(rowp + (((cg >> 1) ^ keyq[q]) << 4) + 8 * (cg & 1)) =
              make_uint2(pack_f16x2(v[0], v[1], f16_sat), pack_f16x2(v[2], v[3], f16_sat));
        } else if constexpr (SPLIT) {
          // v = hi + lo with hi = bf16(v), lo = bf16(v - hi); logical row layout [32 hi | 32 lo]
          const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2));
          const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2], v[3]}, bf16x2));
          const f32x2 r01 = {v[0] - __builtin_bit_cast(float, h01 << 16), v[1] - __builtin_bit_cast(float, h01 & 0xffff0000u)};
          const f32x2 r23 = {v[2] - __builtin_bit_cast(float, h23 << 16), v[3] - __builtin_bit_cast(float, h23 & 0xffff0000u)};
          const unsigned l01 = __builtin_bit_cast(unsigned, __builtin_convertvector(r01, bf16x2));
          const unsigned l23 = __builtin_bit_cast(unsigned, __builtin_convertvector(r23, bf16x2));
          // hi of channels 4cg..4cg+3: piece cg>>1, half cg&1; lo: piece 4 + (cg>>1)
          const int half = 8 * (cg & 1);
          *reinterpret_cast<uint2*>(rowp + (((cg >> 1) ^ keyq[q]) << 4) + half) = make_uint2(h01, h23);
          *reinterpret_cast<uint2*>(rowp + ((((cg >> 1) + 4) ^ keyq[q]) << 4) + half) = make_uint2(l01, l23);
        } else {
          *reinterpret_cast<f32x4*>(rowp + ((cg ^ keyq[q]) << 4)) = v;
        }
      }
    if constexpr (SPLIT && HI) report_f16_saturation(f16_sat_bits_bad(f16_sat), p.flags);
  };

  // `tap` = ConvStage::poff entry: patch row offset of the tap in bits 0..15, its column shift in 16..23, row shift in 24..31
  auto compute = [&](const BGroup<WNB>& RG, int src, int tap) __attribute__((always_inline)) {
    const BFrag& R = RG.f[0];
    const char* base[WM];
    int key[WM];
    const int toff = tap & 0xffff, dpj = (tap >> 16) & 0xff, kdi = hTW * (tap >> 24);
#pragma unroll
    for (int a = 0; a < WM; ++a) {
      const int row = (ABL & 8) ? arow[a] : arow[a] + toff;
      base[a] = (ABL & 8) ? lds + row * CROW : lds + src + row * CROW;
      // key of pixel (li + dpi, lj + dpj): ((lj + dpj) >> 1) = (lj >> 1) + (((lj & 1) + dpj) >> 1)
      if constexpr (SWZ2D) key[a] = (((ak0[a] & 0xffff) + (((ak0[a] >> 16) + dpj) >> 1) + kdi) & 7) << 4;
      else key[a] = (ABL & 8) ? 0 : ((row >> 1) & 7) << 4;
    }
    if constexpr (SPLIT && HI) {
      // 16-bit mode: fp16 operands, one MFMA per product.  32-channel stages: channels 0..31 in pieces 0..3, fragments
      // f[0], f[2]; 64-channel stages (H64): pieces 0..7, fragments f[0..3] = k 0..15, 16..31, 32..47, 48..63
#pragma unroll
      for (int s = 0; s < (H64 ? 4 : 2); ++s) {
        f16x8 ah[WM];
#pragma unroll
        for (int a = 0; a < WM; ++a) ah[a] = *reinterpret_cast<const f16x8*>(base[a] + ((32 * s + 16 * lh) ^ key[a]));
#pragma unroll
        for (int b = 0; b < WNB; ++b) {
          const f16x8 bh = __builtin_bit_cast(f16x8, RG.f[b].f[H64 ? s : 2 * s]);
#pragma unroll
          for (int a = 0; a < WM; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah[a], acc[a][b], 0, 0, 0);
        }
      }
    } else if constexpr (SPLIT) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, R.f[2 * s]);
        const bf16x8 bl = __builtin_bit_cast(bf16x8, R.f[2 * s + 1]);
        bf16x8 ah[WM], al[WM];
#pragma unroll
        for (int a = 0; a < WM; ++a) {
          if constexpr (ABL & 16) {
            ah[a] = __builtin_bit_cast(bf16x8, R.f[a & 3]);
            al[a] = __builtin_bit_cast(bf16x8, R.f[(a + 1) & 3]);
          } else {
            ah[a] = *reinterpret_cast<const bf16x8*>(base[a] + ((32 * s + 16 * lh) ^ key[a]));
            al[a] = *reinterpret_cast<const bf16x8*>(base[a] + ((64 + 32 * s + 16 * lh) ^ key[a]));
          }
        }
        // D = W (A operand: rows = couts) x patch (B operand: columns = pixels): a lane ends up with ONE pixel and
        // four runs of 4 consecutive couts, i.e. 16-byte pieces of the pixel's row (conv_epilogue.h).  Small cross
        // terms first, the dominant hi*hi product last; consecutive MFMAs hit different accumulators.
#pragma unroll
        for (int a = 0; a < WM; ++a) acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, ah[a], acc[a][0], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < WM; ++a) acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, al[a], acc[a][0], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < WM; ++a) acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah[a], acc[a][0], 0, 0, 0);
      }
    } else {
      // 32x32x2 fp32 MFMA: lane l supplies k = l>>5; a lane reads ONE float4 at channel 8g + 4*(l>>5)
      // and feeds its components to 4 consecutive MFMAs (K order (0,4),(1,5),(2,6),(3,7) on both operands).
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 fb = R.f[g];
        f32x4 fa[WM];
#pragma unroll
        for (int a = 0; a < WM; ++a) fa[a] = *reinterpret_cast<const f32x4*>(base[a] + ((32 * g + 16 * lh) ^ key[a]));
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int a = 0; a < WM; ++a)
            acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[e], fa[a][e], acc[a][0], 0, 0, 0);
      }
    }
  };

  // ---- tap loop ----------------------------------------------------------------------------------
  // ONE loop over the flattened tap sequence of the launch (stage 0 tap 0, tap 1, ..., stage 1 tap 0, ...),
  // unrolled by the depth of the weight ring so that tap g always computes from ring group g % RING and
  // fetches tap g + RING-1 into group (g + RING-1) % RING: the accumulators and the ring flow through one
  // straight chain (a per-tap-count switch of bodies made the compiler copy all accumulators at every merge).
  // Everything that depends on the position inside the stage (t of NT taps) is a uniform scalar branch:
  //   t == 0        barrier: patch `st` is visible and every wave is done with the buffer patch st+1 overwrites;
  //   t == TP       the next stage's patch is requested right after the fetch of the stage's LAST tap
  //                 (TP = NT-1-AHEAD), so every weight wait of the stage is for a load issued before the patch
  //                 request (vmcnt retires in order) and the patch stays in flight until the end of the stage;
  //   t >= AHEAD    wait for this tap's weights: younger operations = WL * AHEAD weight loads (+ CNQ once the
  //                 patch has been requested, t >= TP); the first AHEAD taps of a stage landed at the previous
  //                 stage's end;
  //   t == NT-1     everything in flight (next taps' weights, the patch) lands before the transform / barrier.
  constexpr int AHEAD = RING - 1;
  constexpr int WL = (HI32 ? 2 : 4) * WNB;  // weight loads per tap and wave
  BGroup<WNB> R[RING < 3 ? 3 : RING] = {};
  KCONV_TS(1);
  const int nstages = st_hi - st_lo;  // stages of this block
  const int last = nstages - 1;
  // Cursor state lives in scalar registers; table fields are re-read only when a cursor enters a new stage, and the
  // tap descriptor of the next step is loaded at the end of the current one (no scalar-load latency inside a step).
  int st = 0, t = 0;                       // current tap: stage st, tap t
  int NT = stages[0].ntaps;                // taps of the current stage
  int tap = stages[0].poff[0];             // ConvStage::poff entry of the current tap
  int fs = 0, ft = 0;                      // fetch cursor: AHEAD taps further on
  int fnt = stages[0].ntaps;               // (clamped to the last stage: refetched, never consumed)
  int64_t fstride = stages[0].tap_stride;
  const float* fw = (const float*)stages[0].wt;
  auto fetch = [&](BGroup<WNB>& R) __attribute__((always_inline)) {
    if constexpr (ABL & 2) {
    } else if constexpr (HI32) {
      load_b_asm_hi(R.f[0], fw, nb_off);
    } else {
      load_b_asm(R.f[0], fw, nb_off);
      if constexpr (WNB == 2) load_b_asm(R.f[1], fw, nb_off + 4096u);
    }
    if (++ft >= fnt) {
      ft = 0;
      fs = fs < last ? fs + 1 : last;
      const ConvStage VFX_CONST& F = stages[fs];
      fw = (const float*)F.wt;
      fstride = F.tap_stride;
      fnt = F.ntaps;
    } else {
      fw += fstride;
    }
  };
  auto use_group = [&](BGroup<WNB>& R) __attribute__((always_inline)) {
    if constexpr (HI32) {
      use_b_hi(R.f[0]);
    } else {
      use_b(R.f[0]);
      if constexpr (WNB == 2) use_b(R.f[1]);
    }
  };
  auto step = [&](BGroup<WNB>& cur_r, BGroup<WNB>& fetch_r) __attribute__((always_inline)) {
    const int TP = NT > AHEAD ? NT - 1 - AHEAD : 0;
    const int cur = (st & 1) * CPATCH, nxt = CPATCH - cur;
    KCONV_TAP_A();
    if (t == 0 && !(ABL & 4)) __syncthreads();
    fetch(fetch_r);
    // The last stage of a block requests nothing (round 6; it used to refetch its own patch and transform it again so that the
    // waits below could count on CNQ loads in flight: phase stamps of the deep ResUNet levels, profiles/r06_c66_conv_timing.txt,
    // put 1.7 us of patch latency + 1.3 us of transform on every stage of blocks that live 23-25 us)
#ifdef VFX_LAST_STAGE_REFETCH  // (measurement builds: the form of rounds 1-5)
    const bool more = true;
#else
    const bool more = st < last;
#endif
    if constexpr (!(ABL & 1))
      if (t == TP && more) issue_patch(stages[st < last ? st + 1 : last], nxt);
    // The waits carry no register operands on purpose: a conditional asm that redefined the ring group would end
    // in a merge, and a merge copy placed before the wait would read registers whose load is still in flight.  The
    // group becomes readable at the unconditional use_b() below.
    if (t >= AHEAD && !(ABL & 3)) {
      if (t >= TP && more) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(WL * AHEAD + CNQ) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(WL * AHEAD) : "memory");
    }
    KCONV_TAP(0);
    use_group(cur_r);
    compute(cur_r, cur, tap);
    __builtin_amdgcn_sched_barrier(0);  // keep the fragment reads of the next tap below the MFMAs of this one (register pressure)
    KCONV_TAP(1);
    if (t == NT - 1) {
      if constexpr (ABL & 64) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(WL * AHEAD + CNQ) : "memory");  // patch latency never exposed
      else asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (praw && more) transform_patch(stages[st < last ? st + 1 : last], nxt);
      ++st;
      t = 0;
      NT = stages[st < last ? st : last].ntaps;
    } else {
      ++t;
    }
    tap = stages[st < last ? st : last].poff[t];  // next step's tap (all fragment reads of this step have been consumed)
    KCONV_TAP(2);
  };
  if constexpr (RING <= 3) {
    fetch(R[0]);
    if constexpr (RING == 3) fetch(R[1]);
    issue_patch(stages[0], 0);
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    // Ring groups that no fetch has filled yet are defined by copies made AFTER the wait: a copy of a group whose
    // load is still in flight reads stale registers, and since a copy makes the two groups the same value for the
    // compiler, the stale one may end up feeding the first tap.
    KCONV_TS(2);
    use_group(R[0]);
    if constexpr (RING == 3) {
      use_group(R[1]);
    } else {
      R[1] = R[0];
    }
    R[2] = R[1];
    if (praw) transform_patch(stages[0], 0);
    KCONV_TS(3);
    if constexpr (RING == 3) {
      while (true) {
        step(R[0], R[2]);
        if (st > last) break;
        step(R[1], R[0]);
        if (st > last) break;
        step(R[2], R[1]);
        if (st > last) break;
      }
    } else {
      while (true) {
        step(R[0], R[1]);
        if (st > last) break;
        step(R[1], R[0]);
        if (st > last) break;
      }
    }
  } else {
    // deeper rings (measurement builds, -DVFX_RING32=N): the same scheme for any depth
#pragma unroll
    for (int g = 0; g < AHEAD; ++g) fetch(R[g]);
    issue_patch(stages[0], 0);
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
#pragma unroll
    for (int g = 0; g < AHEAD; ++g) use_group(R[g]);
    R[RING - 1] = R[AHEAD - 1];
    if (praw) transform_patch(stages[0], 0);
    bool more = true;
    while (more) {
#pragma unroll
      for (int g = 0; g < RING; ++g) {
        if (more) {
          step(R[g], R[(g + AHEAD) % RING]);
          more = st <= last;
        }
      }
    }
  }

  KCONV_TS(4);
  // ---- epilogue: bias + residual, raw and / or activated output (conv_epilogue.h) ------------------
  if constexpr (ABL & 32) {
    float keep = 0.f;
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) keep += acc[a][0][r] + acc[a][WNB - 1][r];
    if (keep == 12345.678f) p.out_act[tid] = keep;  // keeps the accumulators live
  } else {
    float* partial = KS > 1 ? p.ws + (int64_t)ks * ((int64_t)p.B * p.out_img_stride * p.Cout) : nullptr;
    conv_epilogue<BN, WM, WNB, WAVES_N, SPLIT, EPI_HALVES, RA>(p, smem, otab, acc, n0, partial);
  }
  KCONV_TS(5);
  KCONV_TS_END();
}

static size_t conv_lds_bytes(int BN, bool hi) {
  const int halves = (BN == 128 && hi) ? 2 : 1;  // as EPI_HALVES in the kernel
  const size_t main_bytes = std::max<size_t>((size_t)2 * CPATCH, (size_t)CBM * (BN / halves + 4) * 4);
  return main_bytes + kOtabSlots * CBM * 4;
}

template <int BN, bool ELU, bool SPLIT, int ABL = 0, int RING = 3, bool HI = false, bool H64 = false, bool RA = false, bool VL = false>
static void launch_one(int grid, hipStream_t stream, const TapConvParams* dparams) {
  const size_t lds = conv_lds_bytes(BN, HI);
  static uint64_t attr_devices = 0;  // one static per instantiation
  if (first_use_on_current_device(attr_devices)) {
    VFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv<BN, ELU, SPLIT, ABL, RING, HI, H64, RA, VL>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
#ifdef VFX_TIMING
  static unsigned long long* const timing = getenv("VFX_CONV_TIMING_PTR") ? reinterpret_cast<unsigned long long*>(strtoull(getenv("VFX_CONV_TIMING_PTR"), nullptr, 0)) : nullptr;
  hipLaunchKernelGGL((k_conv<BN, ELU, SPLIT, ABL, RING, HI, H64, RA, VL>), dim3(grid), dim3(256), lds, stream, dparams, timing);
#else
  hipLaunchKernelGGL((k_conv<BN, ELU, SPLIT, ABL, RING, HI, H64, RA, VL>), dim3(grid), dim3(256), lds, stream, dparams);
#endif
}

#ifdef VFX_ABLATION_BUILD
static bool launch_ablated(int abl, int grid, hipStream_t stream, const TapConvParams* dparams) {
  switch (abl) {
    case 1: launch_one<128, false, true, 1>(grid, stream, dparams); return true;
    case 2: launch_one<128, false, true, 2>(grid, stream, dparams); return true;
    case 4: launch_one<128, false, true, 4>(grid, stream, dparams); return true;
    case 8: launch_one<128, false, true, 8>(grid, stream, dparams); return true;
    case 16: launch_one<128, false, true, 16>(grid, stream, dparams); return true;
    case 32: launch_one<128, false, true, 32>(grid, stream, dparams); return true;
    case 7: launch_one<128, false, true, 7>(grid, stream, dparams); return true;
    case 23: launch_one<128, false, true, 23>(grid, stream, dparams); return true;
    case 55: launch_one<128, false, true, 55>(grid, stream, dparams); return true;
    case 64: launch_one<128, false, true, 64>(grid, stream, dparams); return true;
    case 68: launch_one<128, false, true, 68>(grid, stream, dparams); return true;
    case 96: launch_one<128, false, true, 96>(grid, stream, dparams); return true;
    default: return false;
  }
}
// the same for the 16-bit launches on activated fp16 sources (H64: the C = 512 stack, the wide upsamplers, the condnet)
template <bool RA>
static bool launch_ablated_h64(int abl, int grid, hipStream_t stream, const TapConvParams* dparams) {
  switch (abl) {
    case 1: launch_one<128, false, true, 1, 2, true, true, RA>(grid, stream, dparams); return true;
    case 2: launch_one<128, false, true, 2, 2, true, true, RA>(grid, stream, dparams); return true;
    case 4: launch_one<128, false, true, 4, 2, true, true, RA>(grid, stream, dparams); return true;
    case 7: launch_one<128, false, true, 7, 2, true, true, RA>(grid, stream, dparams); return true;
    case 16: launch_one<128, false, true, 16, 2, true, true, RA>(grid, stream, dparams); return true;
    case 32: launch_one<128, false, true, 32, 2, true, true, RA>(grid, stream, dparams); return true;
    case 64: launch_one<128, false, true, 64, 2, true, true, RA>(grid, stream, dparams); return true;
    case 68: launch_one<128, false, true, 68, 2, true, true, RA>(grid, stream, dparams); return true;
    default: return false;
  }
}
#endif

template <bool ELU, bool SPLIT, bool HI = false, bool H64 = false, bool RA = false, bool VL = false>
static void launch_bn(int BN, int grid, hipStream_t stream, const TapConvParams* dparams) {
  switch (BN) {
    // H64: a tap is four K = 16 steps (as long as two taps of the 32-channel form), so one tap of look-ahead covers the
    // same time with a third less ring registers (with three groups the BN = 128 tile spills at three waves per SIMD)
    case 128: launch_one<128, ELU, SPLIT, 0, H64 ? 2 : 3, HI, H64, RA, VL>(grid, stream, dparams); break;
    case 64: launch_one<64, ELU, SPLIT, 0, H64 ? 2 : 3, HI, H64, RA, VL>(grid, stream, dparams); break;
#ifndef VFX_RING32
#define VFX_RING32 2
#endif
    default: launch_one<32, ELU, SPLIT, 0, VFX_RING32, HI, H64, RA, VL>(grid, stream, dparams); break;  // 6-MFMA taps: the third group only costs registers (A/B on one box: -4 %)
  }
}
// the same with the per-clip lengths of a varlen batch (TapConvParams::lens): the VL variants exist without an ELU prologue only
// (the vocoder applies its ELUs in the producers' epilogues)
template <bool SPLIT, bool HI = false, bool H64 = false, bool RA = false>
static void launch_vl(bool vl, bool elu, int BN, int grid, hipStream_t stream, const TapConvParams* dparams) {
  if (vl) {
    VFX_CHECK(!elu, "conv: a launch with per-clip lengths cannot have an ELU prologue");
    launch_bn<false, SPLIT, HI, H64, RA, true>(BN, grid, stream, dparams);
  } else if (elu) {
    if constexpr (!H64) launch_bn<true, SPLIT, HI, H64, RA>(BN, grid, stream, dparams);
  } else {
    launch_bn<false, SPLIT, HI, H64, RA>(BN, grid, stream, dparams);
  }
}

// Couts per block: the widest tile that divides Cout, unless that leaves the chip under-filled (the deep
// ResUNet levels have 16 .. 64 spatial tiles): then narrower tiles, i.e. more blocks of less work each.
int conv_block_n(const TapConvParams& hp) {
  const int64_t spatial = (int64_t)hp.B * hp.tiles_h * hp.tiles_w;
  // phased launch: a block covers a whole number of phases or a part of one (a wave's 32 couts never straddle two: 32 | cout_phase);
  // VFX_TUNE_TWO_LAUNCH_UPSAMPLERS: at most one phase per block (the form of rounds 2-5)
  const bool one_phase = hp.nphase > 1 && (hp.tuning & VFX_TUNE_TWO_LAUNCH_UPSAMPLERS);
  auto fits = [&](int bn) {
    if (hp.Cout % bn) return false;
    if (hp.nphase <= 1) return true;
    return hp.cout_phase % bn == 0 || (!one_phase && bn % hp.cout_phase == 0);
  };
  int bn = fits(128) ? 128 : (fits(64) ? 64 : 32);
  // (round 6, r06c65: 768 instead of 384 here, or level 5 of a 10-s batch without split-K on either tile: all slower, profiles/r06_c64_*)
  // ... but never past ONE round of the chip's block slots once the K slices are counted (round 6: a group of ~13 clips of a mixed-length
  // call ran level 5 as 26 tiles x 12 cout blocks x 4 slices = 1 248 blocks of k_conv<32>, two rounds of blocks that live ~24 us
  // whatever they compute; 624 blocks of k_conv<64> are one round).  Batches of 16 x 10 s choose as before.
  const int64_t ks = hp.ksplit > 1 ? hp.ksplit : 1;
#ifdef VFX_BN_RULE_OLD  // (measurement builds)
  while (bn > 32 && spatial * (hp.Cout / bn) < 384) bn >>= 1;
#else
  while (bn > 32 && spatial * (hp.Cout / bn) < 384 && spatial * (hp.Cout / (bn / 2)) * ks <= 768) bn >>= 1;
#endif
  return bn;
}

void launch_conv(const TapConvParams& hp, const TapConvParams* dparams, hipStream_t stream) {
  if (hp.up16) {  // a ConvTranspose1d of the 16-bit mode on its own kernel (upsample16.hip)
    launch_upsample16(hp, dparams, stream);
    return;
  }
  VFX_CHECK(hp.nstages > 0 && hp.P > 0 && hp.P <= kPatchMaxRows && hp.TH * hp.TW <= CBM, "conv: bad stage geometry");
  VFX_CHECK(hp.Cout % 32 == 0, "conv: Cout=%d is not a multiple of 32", hp.Cout);
  bool elu = false;
  for (int s = 0; s < hp.nseg; ++s) elu = elu || hp.seg[s].act == ACT_ELU;
  if (elu)
    for (int s = 0; s < hp.nseg; ++s)
      VFX_CHECK(hp.seg[s].act == ACT_ELU, "conv: ELU cannot be mixed with other prologues in one launch");
  const int BN = conv_block_n(hp);
  const int KS = hp.ksplit > 1 ? hp.ksplit : 1;
  VFX_CHECK(KS == 1 || (hp.ws && hp.nphase <= 1 && !hp.hionly && KS <= hp.nstages), "conv: bad split-K launch");
  const int64_t grid = (int64_t)hp.B * hp.tiles_h * hp.tiles_w * (hp.Cout / BN) * KS;
  VFX_CHECK(grid > 0 && grid < ((int64_t)1 << 31), "conv: bad grid");
#ifdef VFX_ABLATION_BUILD
  static const int abl = getenv("VFX_ABLATE") ? atoi(getenv("VFX_ABLATE")) : 0;
  if (abl && hp.split && !elu && BN == 128 && launch_ablated(abl, (int)grid, stream, dparams)) return;
#endif
  const bool vl = hp.lens != nullptr;
  VFX_CHECK(!vl || KS == 1, "conv: a launch with per-clip lengths cannot be split along K");
  if (hp.split && hp.hionly) {
    // 16-bit mode: activated sources are fp16 tensors with 64-channel stages, raw fp32 sources keep 32-channel stages;
    // one launch has one kind (the hand-counted weight waits depend on the loads per tap)
    int n_act = 0;
    for (int s = 0; s < hp.nseg; ++s) n_act += hp.seg[s].src_act ? 1 : 0;
    VFX_CHECK(n_act == 0 || n_act == hp.nseg, "conv: 16-bit launches cannot mix activated and raw sources");
    VFX_CHECK(!hp.residual_act || n_act, "conv: an activated residual goes with activated sources");
    if (n_act) {
      VFX_CHECK(!elu, "conv: an activated source has no prologue");
#ifdef VFX_ABLATION_BUILD
      if (abl && BN == 128 && !vl && (hp.residual_act ? launch_ablated_h64<true>(abl, (int)grid, stream, dparams) : launch_ablated_h64<false>(abl, (int)grid, stream, dparams))) return;
#endif
      if (hp.residual_act) launch_vl<true, true, true, true>(vl, false, BN, (int)grid, stream, dparams);
      else launch_vl<true, true, true>(vl, false, BN, (int)grid, stream, dparams);
    } else {
      launch_vl<true, true>(vl, elu, BN, (int)grid, stream, dparams);
    }
  } else if (hp.split) {
    launch_vl<true>(vl, elu, BN, (int)grid, stream, dparams);
  } else {
    launch_vl<false>(vl, elu, BN, (int)grid, stream, dparams);
  }
  VFX_HIP(hipGetLastError());
}

// Split-K decision: depends on the geometry of ONE clip (never on B), so that a clip's result does not depend on the batch it
// is restored in.  Parallelism of a clip = spatial tiles x 32-cout blocks; below ~128 the launch is cut into up to 8
// stage ranges of at least 3 stages (27 taps of a 3x3 convolution) each.
int choose_ksplit(const TapConvParams& p) {
  if (p.hionly || p.nphase > 1 || p.per_tap || (p.tuning & VFX_TUNE_NO_SPLITK)) return 1;
  // the reduce pass rewrites the whole output tensor: only launches that own all of it (not the parity classes of a
  // transposed convolution, whose launches interleave their pixels)
  if (p.sh != 1 || p.sw != 1 || p.oh0 != 0 || p.ow0 != 0 || p.Hg != p.Ho || p.Wg != p.Wo) return 1;
  const int par = p.tiles_h * p.tiles_w * (p.Cout / 32);
  // at most 8 slices; a batch of ordinary clips: >= 3 stages per slice, 128 blocks per clip aimed at (round 2); a short clip
  // (TapConvParams::short_clip: <= 128 padded frames): slices of one stage, 512 blocks per clip -- measured both ways in round 4
  // (profiles/r04_c8_splitk_ab.txt: each rule loses 20 % on the other's workload)
  // a LONG clip (short_clip < 0: -(padded frames / 1024), set by the ResUNet plans; round 5) comes in small batches by nature -- 6 clips
  // of 60 s are all the tensor addressing allows -- so its deep launches aim at proportionally more blocks per clip: a 60-s
  // segment (the handler's unit) ran its level-6 / bottleneck convolutions as 72 blocks walking 108 stages each, 0.13 - 0.36 ms
  // per launch against 0.05 ms for the same level of a 16 x 10 s batch (profiles/r05_1x60_vs_16x10_per_launch.txt)
  const int min_stages = p.short_clip > 0 ? 1 : 3;
  const int aim = p.short_clip > 0 ? 512 : (p.short_clip < 0 ? std::min(1024, 128 * -p.short_clip) : 128);
  int s = std::min(8, std::min(p.nstages / min_stages, aim / std::max(par, 1)));
  int pow2 = 1;
  while (pow2 * 2 <= s) pow2 *= 2;
  return pow2;
}

// Sum of the split-K slices (in slice order) + the epilogue of conv_epilogue.h: bias, residual, raw output and / or the
// activated output for the consumer convolution.  One thread = 8 consecutive channels of one pixel (in split-bf16 mode
// exactly the 16-byte hi block and the 16-byte lo block of an 8-channel group).
template <bool SPLIT>
__global__ __launch_bounds__(256) void k_splitk_reduce(const float* __restrict__ ws, int KS, int64_t slice, int64_t ngroups, int Cout,
                                                        const float* __restrict__ bias, const float* __restrict__ residual,
                                                        float* __restrict__ out, float* __restrict__ out_act,
                                                        const float* __restrict__ asc, const float* __restrict__ ash, float slope,
                                                        int elu) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= ngroups) return;
  const int gpp = Cout >> 3;  // 8-channel groups per pixel
  const int64_t pix = idx / gpp;
  const int ch = (int)(idx - pix * gpp) * 8;
  const int64_t e = pix * Cout + ch;
  f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < KS; ++s) {
    v0 += *(const VFX_GLOBAL f32x4*)(ws + s * slice + e);
    v1 += *(const VFX_GLOBAL f32x4*)(ws + s * slice + e + 4);
  }
  if (bias) {
    v0 += *(const VFX_GLOBAL f32x4*)(bias + ch);
    v1 += *(const VFX_GLOBAL f32x4*)(bias + ch + 4);
  }
  if (residual) {
    v0 += *(const VFX_GLOBAL f32x4*)(residual + e);
    v1 += *(const VFX_GLOBAL f32x4*)(residual + e + 4);
  }
  if (out) {
    *(VFX_GLOBAL f32x4*)(out + e) = v0;
    *(VFX_GLOBAL f32x4*)(out + e + 4) = v1;
  }
  if (out_act) {
    float u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float x = i < 4 ? v0[i] : v1[i - 4];
      const float t = x * (asc ? asc[ch + i] : 1.f) + (ash ? ash[ch + i] : 0.f);
      u[i] = elu ? (t > 0.f ? t : expm1f(t)) : fmaxf(t, t * slope);
    }
    if constexpr (SPLIT) {
      unsigned hi[4], lo[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{u[2 * i], u[2 * i + 1]}, bf16x2));
        const f32x2 r = {u[2 * i] - __builtin_bit_cast(float, h << 16), u[2 * i + 1] - __builtin_bit_cast(float, h & 0xffff0000u)};
        hi[i] = h;
        lo[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
      }
      // operand form of a 32-channel chunk: [32 hi bf16 | 32 lo bf16]; this group's 8 hi values are 4 floats at g * 4
      float* base = out_act + pix * Cout + (ch & ~31) + ((ch & 31) >> 3) * 4;
      *(VFX_GLOBAL u32x4*)(base) = u32x4{hi[0], hi[1], hi[2], hi[3]};
      *(VFX_GLOBAL u32x4*)(base + 16) = u32x4{lo[0], lo[1], lo[2], lo[3]};
    } else {
      *(VFX_GLOBAL f32x4*)(out_act + e) = f32x4{u[0], u[1], u[2], u[3]};
      *(VFX_GLOBAL f32x4*)(out_act + e + 4) = f32x4{u[4], u[5], u[6], u[7]};
    }
  }
}

void launch_splitk_reduce(const TapConvParams& hp, hipStream_t stream) {
  const int64_t npix = (int64_t)hp.B * hp.out_img_stride;
  const int64_t ngroups = npix * (hp.Cout / 8);
  const int64_t slice = npix * hp.Cout;
  const unsigned grid = (unsigned)((ngroups + 255) / 256);
  if (hp.split)
    hipLaunchKernelGGL(k_splitk_reduce<true>, dim3(grid), dim3(256), 0, stream, hp.ws, hp.ksplit, slice, ngroups, hp.Cout, hp.bias,
                       hp.residual, hp.out, hp.out_act, hp.act_scale, hp.act_shift, hp.act_slope, hp.act_elu);
  else
    hipLaunchKernelGGL(k_splitk_reduce<false>, dim3(grid), dim3(256), 0, stream, hp.ws, hp.ksplit, slice, ngroups, hp.Cout, hp.bias,
                       hp.residual, hp.out, hp.out_act, hp.act_scale, hp.act_shift, hp.act_slope, hp.act_elu);
  VFX_HIP(hipGetLastError());
}

double conv_flops(const TapConvParams& hp) {
  if (hp.flops_override > 0) return hp.flops_override;
  double k = 0;
  for (int s = 0; s < hp.nseg; ++s) k += (double)hp.seg[s].ntaps * hp.seg[s].C;
  return 2.0 * (double)hp.M * hp.Cout * k;
}

}  // namespace vfx
