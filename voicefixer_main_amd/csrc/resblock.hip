// resblock.hip -- one fused TFGAN ResStack layer (oracle/vocoder.py, layer table in vfx_config):
//
//     y = x + conv2(LeakyReLU(conv1(LeakyReLU(x)) + b1)) + b2      conv1: k3, dilation d;  conv2: k3, dilation 1
//
// for the channel counts whose unfused form is HBM-bound (C = 64, 128; 44.1 kHz and 14.7 kHz stacks).
// The two convolutions of a layer move 20 bytes per element through HBM as separate launches (x, h
// written and re-read, the residual); fused, h never leaves the CU: 8 bytes per element.
//
// Tile = up to 128 positions of h (= conv1 output incl. the +-1 halo conv2 needs) of ONE clip:
//   * d <= 32: 128 consecutive positions, 126 outputs; x patch = 128 + 2d consecutive positions;
//   * d  > 32: "folded" -- the sequence is viewed as rows of d samples, conv1's taps become vertical
//     neighbours: h tile = TH x (TW + 2) positions (7 x 18), outputs TH x TW, x patch (TH+2) x (TW+2).
//     A position is just row*d + col, so a column index of -1 or d is simply the neighbouring row.
// Phases (one block = 4 or 8 waves, wave = (128 / WAVES_M) h rows x 32 channels, all C output channels per block):
//   1. conv1: per 32-channel chunk the raw x patch arrives by LDS-DMA (zero fill by the buffer bound),
//      is turned into MFMA operand form in place (LeakyReLU, hi/lo split, swizzled slots) and read by the
//      three taps -- same machinery as k_conv (conv.hip), statically scheduled here;
//   2. h = LeakyReLU(acc + b1), zero outside the sequence, written to LDS in operand form -- over the patch buffers,
//      which are dead by then (one more barrier; 48 / 68 KB of LDS instead of 80 / 112);
//   3. conv2: A fragments straight from the LDS-resident h, no staging, no barriers;
//   4. epilogue: + b2 + x (residual, L2-hot), fp32 rows of 16 bytes per lane.
// Weights as in k_conv: fragment order, global -> VGPR ring, inline-asm loads with hand-counted vmcnt.
// Arithmetic: split-bf16 (precision 1) or, with ResBlockParams::hionly (precision 2), fp16 operands in the hi halves
// only, one MFMA per product; the fp32 mode keeps the two-launch plan.
#include "conv_common.h"
#include "vfx_internal.h"

namespace vfx {

// n is a compile-time constant after unrolling; only these counts occur (NG = patch DMA instructions per wave,
// WL = weight loads per tap and wave)
#ifndef VFX_RB_RING32
#define VFX_RB_RING32 3  // weight ring of the C = 32 2-D block (see RING below); 5 needs three blocks per CU instead of four
#endif

template <int NG, int WL, bool HI>
__device__ __forceinline__ void wait_b_dyn(BFrag& R, int n) {
#define VFX_WAIT_CASE(v) case v: asm volatile("s_waitcnt vmcnt(%0)" : : "n"(v) : "memory"); break;
  switch (n) {  // n is a compile-time constant after unrolling: AHEAD weight fetches (WL loads each) [+ one patch request (NG)]
    VFX_WAIT_CASE(1) VFX_WAIT_CASE(2) VFX_WAIT_CASE(3) VFX_WAIT_CASE(4) VFX_WAIT_CASE(5) VFX_WAIT_CASE(6) VFX_WAIT_CASE(7)
    VFX_WAIT_CASE(8) VFX_WAIT_CASE(9) VFX_WAIT_CASE(10) VFX_WAIT_CASE(11) VFX_WAIT_CASE(12) VFX_WAIT_CASE(13) VFX_WAIT_CASE(14)
    VFX_WAIT_CASE(15) VFX_WAIT_CASE(16) VFX_WAIT_CASE(17) VFX_WAIT_CASE(18) VFX_WAIT_CASE(19) VFX_WAIT_CASE(20) VFX_WAIT_CASE(21)
    VFX_WAIT_CASE(22) VFX_WAIT_CASE(23) VFX_WAIT_CASE(24)
    default: asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); break;
  }
#undef VFX_WAIT_CASE
  if constexpr (HI) use_b_hi(R);
  else use_b(R);
}

// NW waves per block: 4 (C = 64) or 8 (C = 128: h needs all 128 couts of conv1 in one block).
// Waves per SIMD the register budget is cut for: C = 64 (4-wave blocks, 48 KB of LDS): three blocks per CU;
// C = 128 (8-wave blocks, 68 KB): two blocks per CU in the 16-bit mode (116 VGPRs), one in split mode (140).
// G2 = true: the same machinery as a fused 2-D ConvBlockRes of the ResUNets (models/components/modules.py:223-271,
// Cin == Cout, identity shortcut):  y = x + conv2(lrelu(bn2(conv1(lrelu(bn1(x))))))  with 3x3 convolutions.  Tile = h grid
// of TH x W1 = 128 pixels (8 x 16 or 16 x 8), outputs = its interior, x patch = (TH + 2) x (W1 + 2) = 180 pixels.
// MT = 256 (round 4; 2-D mode at C = 32 only: one 32-channel chunk = one patch buffer): h grid of 16 x 16 pixels, 14 x 14 outputs,
// x patch 18 x 18 = 324 pixels -- the halo costs 1.31x recomputed h positions instead of 1.52x (8 x 16 / 16 x 8: 84 outputs per
// 128) and 1.65 instead of 2.14 patch pixels per output.  Four waves of 64 pixels (the 128-position tile: two), 48 KB of LDS:
// three blocks per CU = the same 12 waves per CU as six blocks of the small tile.  Measured (profiles/r04_c13_c64_tile16x16_ab.txt):
// mel ResUNet 13.47 -> 13.23 ms, ssr_sr64 119.4 -> 116.5 ms, stream1s 2.50 -> 2.39 ms.  At C = 64 the two patch buffers are 96 KB,
// i.e. one 8-wave block per CU instead of three 4-wave ones: built as a variant, measured +0.5 %, deleted.
// IN1 = true (round 4; 16 x 16 tiles only): the ENTRY block of a ResUNet, encoder_block1.conv_block1 -- Cin = 1, so conv1 is nine
// multiply-adds per (pixel, channel) and the shortcut one (modules.py:223-271 with a 1x1 shortcut): every thread computes the 32
// channels of one h pixel from an 18 x 18 single-channel patch and writes them to LDS in operand form (phases 1 + 2 without MFMA),
// conv2 and the epilogue are the block's own, the residual is  wsc * x + bsc  of the raw input sample.  One launch reading 4 and
// writing 128 bytes per pixel instead of k_conv_c1 (h and the shortcut out: 256 bytes) + k_conv (both back in, y out: 384).
// SC2 = true (round 4; 16 x 16 tiles, C = 32): the FIRST block of decoder level 1 (decoder_block6.conv_block2: x = cat(upsampled,
// skip), 64 -> 32 channels, 1x1 shortcut with bias).  conv1 runs over the two sources as two 32-channel chunks through the ONE patch
// buffer (a second one would leave one block per CU), conv2 is the block's own, and the shortcut is one more K segment AFTER conv2:
// the raw centre pixels of each source come in again (L2), are split without activation and multiplied by the shortcut's fragment
// group.  12 bytes per element through HBM instead of the 28 of the two k_conv launches (h out and back in, the sources twice).
template <int C, int NW, bool HI, bool G2 = false, int MT = 128, bool IN1 = false, bool SC2 = false>
__global__ __launch_bounds__(NW * 64, MT == 256 ? 3 : (NW == 2 ? 3 : (NW == 4 ? (C == 32 ? (VFX_RB_RING32 >= 5 ? 3 : 4) : 3) : (HI ? 4 : 2)))) void k_resblock(const ResBlockParams* __restrict__ pp) {
  static_assert(MT == 128 || (MT == 256 && G2 && C == 32 && NW == 4 && !HI), "the 256-position tile exists for the C = 32 2-D block");
  static_assert(!IN1 || MT == 256, "the entry block runs 16 x 16 tiles");
  static_assert(!SC2 || (MT == 256 && !IN1), "the two-source block runs 16 x 16 tiles");
  constexpr int IN1_OFF = 40960;  // IN1: the activated 18 x 18 input patch (fp32), behind h (32 KB) and the staged output (37.9 KB)
  constexpr int PMAX = MT + MT / 2;          // patch rows per buffer (192; 384 for the 18 x 18 patch of the 16 x 16 tile)
  constexpr int PBYTES = PMAX * CROW;        // bytes per patch buffer
  constexpr int KT = G2 ? 9 : 3;  // taps per convolution
  constexpr int WL = HI ? 2 : 4;  // weight loads per tap and wave (HI: fp16 operands, hi fragments only)
  constexpr int NTHR = NW * 64;
  constexpr int RG = NTHR / 8;               // patch rows per DMA instruction group (8 lanes per row)
  constexpr int NG = PMAX / RG;     // DMA instructions per wave and patch
  static_assert(PMAX % RG == 0, "patch rows must split into whole DMA groups");
  constexpr int NCH = C / 32;  // 32-channel chunks = waves along N
  constexpr int WAVES_N = NCH, WAVES_M = NW / WAVES_N, WM = (MT / 32) / WAVES_M;  // (C, NW) = (64, 4): 2, 2, 2;  (128, 8): 4, 2, 2
  // weight taps in flight.  C = 32 (2-D blocks): the timing-only build without weight refreshes (-DVFX_RB_ABL_NOWEIGHTS) runs that
  // block 25 % faster, but deeper rings do not (-DVFX_RB_RING32=4: 2.97 ms per step against 2.95 with 3; 5 at three blocks per CU:
  // 3.40): it is not the latency of a fetch but their number -- the four waves of a block (32 pixels x ALL 32 output channels each)
  // fetch the same 74 KB of fragments per tile, four blocks per CU: ~30 B/clk of the CU's 64 B/clk L2 path.
  constexpr int RING = WM >= 4 ? 2 : ((C == 32 && G2) ? VFX_RB_RING32 : 3), AHEAD = RING - 1;
  constexpr int NBUF = (NCH == 4 && !G2) ? 3 : 2;  // patch buffers (chunks in flight / in use)
  constexpr int HROW = C * 4;         // bytes per h row
  constexpr int H_OFF = 0;            // h overlays the patch buffers (dead once conv1 is done): less LDS, more blocks per CU
  constexpr int NT1 = SC2 ? 2 * KT : KT * NCH;  // taps of conv1 (chunk-major; SC2: two source chunks); conv2 has KT * NCH
  constexpr int LDO = C + 4;          // staged output row (floats)
  constexpr int OTAB_OFF = MT * LDO * 4;
  // The residual: the epilogue's re-read of x finds its lines evicted from L2 (the HBM-bound C = 64 stack moves 5.0 GB per
  // layer through the fabric against 3.64 GB of tensors, a fifth of it this re-read).  Where registers allow (C = 64: 3
  // waves per SIMD, 168 VGPRs) every thread keeps the raw values it fetched for the patch (48 registers) and adds them to
  // the staged conv2 result in LDS: x is read from memory once.
  constexpr bool KEEPRES = HI && !G2 && C == 64 && NW == 4;  // (split-bf16 mode: 168 VGPRs are not enough, it keeps the re-read)

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
  VFX_TS_DECL;  // timing builds (-DVFX_TIMING, scripts/phase_timing.py --block2d): s_memtime at the phase boundaries, 2-D blocks
  VFX_TS(0);
  const int tj = tile % p.tiles_w;
  const int ti = (tile / p.tiles_w) % p.tiles_h;
  const int img = tile / (p.tiles_w * p.tiles_h);
  const int T = p.T, d = p.dil, W1 = p.W1, TH = p.TH, PW = p.PW, P = p.P;
  const int rowstride = p.fold ? d : 0;
  const int j0 = tj * p.TWo;  // first output column (folded, 2-D) / position (1-D) of the tile
  const int base_h = p.fold ? ti * TH * d + j0 - 1 : j0 - 1;  // position of h pixel (0, 0)
  const int base_x = base_h - d;                                // position of x patch pixel (0, 0)
  const int Hh = p.H, Ww = p.W;      // 2-D mode: image extent
  const int i0 = ti * (TH - 2);      // 2-D mode: first output row of the tile
  // 1-D layers, batches of clips of unequal length (ResBlockParams::lens): this clip's sequence ends at Tb <= T -- positions past
  // it read as zeros, h is zero there, nothing is stored there; a tile wholly past the end has nothing to do
  const int Tb = (!G2 && p.lens) ? min(T, ((const VFX_GLOBAL int*)p.lens)[__builtin_amdgcn_readfirstlane(img)] * p.lens_mul) : T;
  if (!G2 && base_h + 1 >= Tb) return;
  const float slope = p.slope;

  const int lr = tid >> 3, cg = tid & 7;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int wm = wave_u / WAVES_N, wn = wave_u % WAVES_N;
  const int l31 = lane & 31, lh = lane >> 5;
  const int key_l = (lr >> 1) & 7;
  // row / PW and pixel / W1 as multiply-shift (rows < 512, divisors <= 320: exact, cf. resblock_rw.hip): the per-thread tables below were
  // up to fourteen integer divisions of ~25 VALU instructions each -- a seventh of the block's instructions (round 6: the fused blocks
  // are bound by the instructions they issue, ~6 cycles each per SIMD, like the 1-D layers were)
  const unsigned inv_pw = ((1u << 20) + (unsigned)PW - 1u) / (unsigned)PW, inv_w1 = ((1u << 20) + (unsigned)W1 - 1u) / (unsigned)W1;

  // ---- per-thread tables ------------------------------------------------------------------------------
  // x patch pixel lr + 32q -> byte offset in x (chunk 0) / validity
  unsigned voff[NG];
  unsigned okmask = 0;
  // Swizzle key of this thread's patch rows.  1-D patches: (row >> 1) & 7, the same for every row group.  2-D patches
  // (G2, even width PW): ((pj >> 1) + (W1 / 2) * pi) & 7, the key of k_conv's 2-D tiles -- the bank half of row pi*PW + pj
  // is pj & 1, and the lanes of a ds_read_b128 group (runs of consecutive columns of W1-wide h rows) get distinct
  // (bank half, slot) pairs for every tap shift; the row-linear key gave 47-57 % LDS conflict cycles there (PMC).
  int keyq[NG];
  const int hW1 = W1 >> 1;
#pragma unroll
  for (int q = 0; q < NG; ++q) {
    const int prow = lr + RG * q;
    const int pi = (int)(((unsigned)prow * inv_pw) >> 20), pj = prow - pi * PW;
    keyq[q] = G2 ? (((pj >> 1) + hW1 * pi) & 7) : key_l;
    if constexpr (G2) {  // patch pixel (pi, pj) = image pixel (i0 - 2 + pi, j0 - 2 + pj)
      const int r = i0 - 2 + pi, c = j0 - 2 + pj;
      const bool ok = (prow < P) & ((unsigned)r < (unsigned)Hh) & ((unsigned)c < (unsigned)Ww);
      voff[q] = (unsigned)((img * Hh + r) * Ww + c) * (unsigned)(C * 4);
      okmask |= ok ? (1u << q) : 0u;
    } else {
      const int pos = base_x + pi * rowstride + pj;
      const bool ok = (prow < P) & ((unsigned)pos < (unsigned)Tb);
      voff[q] = (unsigned)(img * T + pos) * (unsigned)(C * 4);
      okmask |= ok ? (1u << q) : 0u;
    }
  }
  int arow1[WM], arow2[WM];  // A row of this lane's h pixel: in the x patch (tap offset added) / in the h buffer
  int kq0[WM], kpar[WM];     // 2-D mode: swizzle key of its patch pixel before the tap shift, and its column parity
  bool hval[WM];             // that h pixel lies inside the tile's h grid and inside the sequence
#pragma unroll
  for (int a = 0; a < WM; ++a) {
    const int ml = (wm * WM + a) * 32 + l31;
    const int li = (int)(((unsigned)ml * inv_w1) >> 20), lj = ml - li * W1;
    arow1[a] = li < TH ? li * PW + lj : 0;
    arow2[a] = ml;
    kq0[a] = li < TH ? (lj >> 1) + hW1 * li : 0;
    kpar[a] = li < TH ? (lj & 1) : 0;
    if constexpr (G2) {  // h pixel (li, lj) = image pixel (i0 - 1 + li, j0 - 1 + lj)
      hval[a] = (li < TH) & ((unsigned)(i0 - 1 + li) < (unsigned)Hh) & ((unsigned)(j0 - 1 + lj) < (unsigned)Ww);
    } else {
      const int pos = base_h + li * rowstride + lj;
      hval[a] = (li < TH) & ((unsigned)pos < (unsigned)Tb);
    }
  }
  const unsigned nb_off = (unsigned)(wn * 1024 + lane * 4) * 4u;
  const int64_t ts = (int64_t)C * kKC;  // floats per tap of a weight tensor
  f32x4 b1v[4];  // bias of conv1 for this lane's accumulator channels: runs wn*32 + 8j + 4lh .. +3 (2-D mode: none)
#pragma unroll
  for (int j = 0; j < 4; ++j)
    b1v[j] = G2 ? f32x4{0.f, 0.f, 0.f, 0.f} : *(const VFX_GLOBAL f32x4*)(p.b1 + wn * 32 + 8 * j + 4 * lh);

  f32x16 acc[WM];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  // ---- x patch: LDS-DMA request and in-place transform (cf. conv.hip) -------------------------------------
  auto issue_patch = [&](int c, int dst) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(SC2 ? (c == 0 ? p.x : p.x2) : p.x + c * kKC), 0,
        (int)(unsigned)((int64_t)p.B * (G2 ? (int64_t)Hh * Ww : (int64_t)T) * C * 4 - (SC2 ? 0 : (int64_t)c * kKC * 4)), 0x00020000);
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      const unsigned o = (okmask & (1u << q)) ? voff[q] + 16u * cg : 0xfffffff0u;
      VFX_LDS void* l = (VFX_LDS void*)(lds + dst + (RG * q + 8 * wave_u) * CROW);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, (int)o, 0, 0, 0);
    }
  };
  const int nq = (P + RG - 1) / RG;
  f32x4 keep[KEEPRES ? NCH : 1][KEEPRES ? NG : 1];  // raw x of this thread's patch pixels, all chunks
  auto transform_patch = [&](int dst, int c) __attribute__((always_inline)) {
    char* row0 = lds + dst + lr * CROW;
    f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};  // 2-D mode: bn1 of this thread's 4 channels
    if constexpr (G2) {
      psc = *(const VFX_GLOBAL f32x4*)(p.sc1 + c * kKC + 4 * cg);
      psh = *(const VFX_GLOBAL f32x4*)(p.sh1 + c * kKC + 4 * cg);
    }
    f32x4 raw[NG];
    unsigned f16_sat = 0;  // 16-bit mode: a value left the fp16 range and was clamped (reported per patch)
#pragma unroll
    for (int q = 0; q < NG; ++q)
      if (MT == 256 ? q * RG < 18 * 18 : (q * RG < MT || q < nq)) {
        raw[q] = *reinterpret_cast<const f32x4*>(row0 + RG * q * CROW + 16 * cg);
        if constexpr (KEEPRES) keep[c][q] = raw[q];
      }
#pragma unroll
    for (int q = 0; q < NG; ++q)
      if (MT == 256 ? q * RG < 18 * 18 : (q * RG < MT || q < nq)) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (G2) {  // affine first, so the zero fill of the DMA is not zero any more: mask after the activation
            const float t = raw[q][e] * psc[e] + psh[e];
            v[e] = (okmask & (1u << q)) ? fmaxf(t, t * slope) : 0.f;
          } else {
            v[e] = fmaxf(raw[q][e], raw[q][e] * slope);  // LeakyReLU(0) = 0: DMA zero fill stays zero
          }
        }
        if constexpr (HI) {  // fp16 in the hi half only
          *reinterpret_cast<uint2*>(row0 + RG * q * CROW + (((cg >> 1) ^ keyq[q]) << 4) + 8 * (cg & 1)) =
              make_uint2(pack_f16x2(v[0], v[1], f16_sat), pack_f16x2(v[2], v[3], f16_sat));
          continue;
        }
        const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2));
        const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2], v[3]}, bf16x2));
        const f32x2 r01 = {v[0] - __builtin_bit_cast(float, h01 << 16), v[1] - __builtin_bit_cast(float, h01 & 0xffff0000u)};
        const f32x2 r23 = {v[2] - __builtin_bit_cast(float, h23 << 16), v[3] - __builtin_bit_cast(float, h23 & 0xffff0000u)};
        const unsigned l01 = __builtin_bit_cast(unsigned, __builtin_convertvector(r01, bf16x2));
        const unsigned l23 = __builtin_bit_cast(unsigned, __builtin_convertvector(r23, bf16x2));
        char* rowp = row0 + RG * q * CROW;
        const int half = 8 * (cg & 1);
        *reinterpret_cast<uint2*>(rowp + (((cg >> 1) ^ keyq[q]) << 4) + half) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(rowp + ((((cg >> 1) + 4) ^ keyq[q]) << 4) + half) = make_uint2(l01, l23);
      }
    if constexpr (HI) report_f16_saturation(f16_sat_bits_bad(f16_sat), p.flags);
  };

  // ---- MFMA step: 32 channels of one tap; A rows `row[a]` of an LDS image with `stride` bytes per row ------
  // `chunk` >= 0: the image is h (rows of NCH 128-byte chunks): chunk c of row r sits at chunk position c ^ (r & 1).  A row
  // stride that is a multiple of 256 bytes puts every row on the same half of the 64 LDS banks, and rows r, r + 1 share
  // their swizzle key: without the parity swap the fragment reads of conv2 are 2-way bank conflicts (measured: a third of
  // the LDS cycles of the kernel).  Patch rows (128-byte stride) alternate halves by themselves: chunk = -1.
  // use_keyov: `keyov` holds the swizzle keys (already << 4) of the rows -- they are not the row-linear ones (2-D patches).
  auto mma = [&](const BFrag& R, const char* img_base, int stride, const int (&row)[WM], int chunk, const int (&keyov)[WM],
                 bool use_keyov) __attribute__((always_inline)) {
    const char* base[WM];
    int key[WM];
#pragma unroll
    for (int a = 0; a < WM; ++a) {
      base[a] = img_base + row[a] * stride;
      if (chunk >= 0) base[a] += (NCH > 1 ? (chunk ^ (row[a] & 1)) : chunk) * CROW;
      key[a] = use_keyov ? keyov[a] : swz_key(row[a]);
    }
    if constexpr (HI) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const f16x8 bh = __builtin_bit_cast(f16x8, R.f[2 * s]);
        f16x8 ah[WM];
#pragma unroll
        for (int a = 0; a < WM; ++a) ah[a] = *reinterpret_cast<const f16x8*>(base[a] + ((32 * s + 16 * lh) ^ key[a]));
#pragma unroll
        for (int a = 0; a < WM; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah[a], acc[a], 0, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8 bh = __builtin_bit_cast(bf16x8, R.f[2 * s]);
      const bf16x8 bl = __builtin_bit_cast(bf16x8, R.f[2 * s + 1]);
      bf16x8 ah[WM], al[WM];
#pragma unroll
      for (int a = 0; a < WM; ++a) {
        ah[a] = *reinterpret_cast<const bf16x8*>(base[a] + ((32 * s + 16 * lh) ^ key[a]));
        al[a] = *reinterpret_cast<const bf16x8*>(base[a] + ((64 + 32 * s + 16 * lh) ^ key[a]));
      }
      // D = W (A operand: rows = couts) x image rows (B operand: columns = pixels): lane = pixel, registers =
      // four runs of 4 consecutive couts
#pragma unroll
      for (int a = 0; a < WM; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, ah[a], acc[a], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < WM; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, al[a], acc[a], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < WM; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah[a], acc[a], 0, 0, 0);
    }
  };

  // ---- weight ring: global tap g (conv1: 0 .. NT1-1, conv2: NT1 .. 2*NT1-1) in register group g % RING ------
  BFrag R0 = {}, R1 = {}, R2 = {}, R3 = {}, R4 = {};
  auto ring = [&](int g) __attribute__((always_inline)) -> BFrag& {
    const int i = g % RING;
    return i == 0 ? R0 : (i == 1 ? R1 : (i == 2 ? R2 : (i == 3 ? R3 : R4)));
  };
  auto fetch = [&](int g) __attribute__((always_inline)) {
    const float* w = g < NT1 ? p.w1 + g * ts : (g < 2 * NT1 ? p.w2 + (g - NT1) * ts : p.w2 + (NT1 - 1) * ts);
    if constexpr (SC2)  // conv1 of source 0, of source 1, conv2, the two shortcut groups
      w = g < KT ? p.w1 + g * ts
                 : (g < 2 * KT ? p.w1x2 + (g - KT) * ts : (g < 3 * KT ? p.w2 + (g - 2 * KT) * ts : (g == 3 * KT ? p.wsc : p.wsc2)));
#ifdef VFX_RB_ABL_NOWEIGHTS  // timing-only build (wrong results): the weight ring is filled once and never refreshed
    if (g >= RING) return;
#endif
    if constexpr (HI) load_b_asm_hi(ring(g), w, nb_off);
    else load_b_asm(ring(g), w, nb_off);
  };
  auto drain = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    if constexpr (HI) {
      use_b_hi(R0);
      use_b_hi(R1);
      if constexpr (RING >= 3) use_b_hi(R2);
      if constexpr (RING >= 4) use_b_hi(R3);
      if constexpr (RING >= 5) use_b_hi(R4);
    } else {
      use_b(R0);
      use_b(R1);
      if constexpr (RING >= 3) use_b(R2);
      if constexpr (RING >= 4) use_b(R3);
      if constexpr (RING >= 5) use_b(R4);
    }
  };

  // ---- phase 1: conv1 ------------------------------------------------------------------------------------
  // Per chunk: taps 3c .. 3c+2.  Iteration g: fetch tap g + AHEAD; in iteration 3c + 2 - AHEAD (right after the
  // fetch of the chunk's last tap) request the next chunk's patch; wait for tap g; MFMAs.  Every chunk ends
  // with vmcnt(0), so only fetches issued inside the chunk need a counted wait: they all precede the patch
  // request, which is therefore never drained early.
  // (Requesting both patches of a two-chunk block up front was measured: -3 % on the C = 64 stack -- the stack moves
  // ~5 GB per layer through the fabric at ~4.9 TB/s, it is bandwidth-, not latency-bound.)
  if constexpr (IN1) {
    // conv2's first taps are on their way while conv1 is computed
#pragma unroll
    for (int g = 0; g < AHEAD; ++g) fetch(NT1 + g);
    float* pa = reinterpret_cast<float*>(lds + IN1_OFF);
    const float s1 = p.in1_scale, t1 = p.in1_shift;
    for (int e = tid; e < 18 * 18; e += NTHR) {  // patch pixel (pi, pj) = image pixel (i0 - 2 + pi, j0 - 2 + pj)
      const int pi = e / 18, pj = e - pi * 18;
      const int r = i0 - 2 + pi, c = j0 - 2 + pj;
      float v = 0.f;  // zero padding AFTER bn1 + LeakyReLU
      if (((unsigned)r < (unsigned)Hh) & ((unsigned)c < (unsigned)Ww)) {
        const float t = *(const VFX_GLOBAL float*)(p.x + ((int64_t)img * Hh + r) * Ww + c) * s1 + t1;
        v = fmaxf(t, t * slope);
      }
      pa[e] = v;
    }
    __syncthreads();
    {
      const int m = tid, li = m >> 4, lj = m & 15;  // h pixel (li, lj) = image pixel (i0 - 1 + li, j0 - 1 + lj)
      const bool ok = ((unsigned)(i0 - 1 + li) < (unsigned)Hh) & ((unsigned)(j0 - 1 + lj) < (unsigned)Ww);
      float a9[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) a9[k] = pa[(li + k / 3) * 18 + lj + k % 3];
      char* rowp = lds + H_OFF + m * HROW;
      const int key = (m >> 1) & 7;
#pragma unroll
      for (int pc = 0; pc < 4; ++pc) {  // 8 channels = one 16-byte piece of hi and one of lo halves
        f32x4 u[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int k = 0; k < 9; ++k) {
            const f32x4 w = *(const VFX_GLOBAL f32x4*)(p.w1 + k * 32 + 8 * pc + 4 * hh);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] = fmaf(a9[k], w[e], s[e]);
          }
          const f32x4 hsc = *(const VFX_GLOBAL f32x4*)(p.sc2 + 8 * pc + 4 * hh);
          const f32x4 hsh = *(const VFX_GLOBAL f32x4*)(p.sh2 + 8 * pc + 4 * hh);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float t = s[e] * hsc[e] + hsh[e];
            u[hh][e] = ok ? fmaxf(t, t * slope) : 0.f;
          }
        }
        u32x4 hi, lo;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{u[hh][0], u[hh][1]}, bf16x2));
          const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{u[hh][2], u[hh][3]}, bf16x2));
          const f32x2 r01 = {u[hh][0] - __builtin_bit_cast(float, h01 << 16), u[hh][1] - __builtin_bit_cast(float, h01 & 0xffff0000u)};
          const f32x2 r23 = {u[hh][2] - __builtin_bit_cast(float, h23 << 16), u[hh][3] - __builtin_bit_cast(float, h23 & 0xffff0000u)};
          hi[2 * hh] = h01;
          hi[2 * hh + 1] = h23;
          lo[2 * hh] = __builtin_bit_cast(unsigned, __builtin_convertvector(r01, bf16x2));
          lo[2 * hh + 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(r23, bf16x2));
        }
        *reinterpret_cast<u32x4*>(rowp + ((pc ^ key) << 4)) = hi;
        *reinterpret_cast<u32x4*>(rowp + (((pc + 4) ^ key) << 4)) = lo;
      }
    }
    drain();
    __syncthreads();  // h is complete
  } else {
  if constexpr (SC2) {
#pragma unroll
    for (int g = 0; g < AHEAD; ++g) fetch(g);
    issue_patch(0, 0);
    drain();
    transform_patch(0, 0);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      __syncthreads();  // the patch of source c is visible
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        const int g = KT * c + k;
        fetch(g + AHEAD);
        if (k >= AHEAD) wait_b_dyn<NG, WL, HI>(ring(g), WL * AHEAD);  // the first AHEAD taps of a chunk landed with the drain before it
        int rows[WM], kov[WM];
#pragma unroll
        for (int a = 0; a < WM; ++a) {
          rows[a] = arow1[a] + p.poff9[k];
          int k0 = kq0[a];
          asm volatile("" : "+v"(k0));
          kov[a] = ((k0 + hW1 * (k / 3) + (k % 3 == 2 ? 1 : 0) + (k % 3 == 1 ? kpar[a] : 0)) & 7) << 4;
        }
        mma(ring(g), lds, CROW, rows, -1, kov, true);
        __builtin_amdgcn_sched_barrier(0);
      }
      drain();
      if (c == 0) {
        __syncthreads();  // every wave is done with source 0's patch: source 1 goes into the same buffer
        issue_patch(1, 0);
        drain();
        transform_patch(0, 1);
      }
    }
    __syncthreads();  // every wave is done reading the patch buffer that h overlays
  } else {
#pragma unroll
  for (int g = 0; g < AHEAD; ++g) fetch(g);
  issue_patch(0, 0);
  // NBUF = 3 (four chunks: C = 128): chunks 0, 1, 2 arrive in ONE round trip, chunk 3 goes into chunk 0's buffer once conv1 is
  // done with it -- two exposed memory latencies per tile instead of four (a chunk is 0.2 us of MFMAs, a round trip 2 us)
  if constexpr (NBUF == 3) {
    issue_patch(1, PBYTES);
    issue_patch(2, 2 * PBYTES);
  }
  VFX_TS(1);  // patch requested
  drain();
  VFX_TS(2);  // patch arrived
  transform_patch(0, 0);
  VFX_TS(3);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    // the patch requested while this chunk is computed: the next one, or (three buffers) chunk 3 during chunk 1
    const int dma_chunk = NBUF == 3 ? (c == 1 ? 3 : NCH) : c + 1;
    const bool has_dma = dma_chunk < NCH;
    __syncthreads();  // patch c is visible; the buffer of the chunk before it is free
    if (c == 0) VFX_TS(4);
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const int g = KT * c + k;
      fetch(g + AHEAD);
      // the patch is requested right after the fetch of this chunk's LAST tap, so every counted wait
      // below is for a fetch issued before it
      if (k == KT - 1 - AHEAD && has_dma) issue_patch(dma_chunk, (dma_chunk % NBUF) * PBYTES);
      if (k >= AHEAD) wait_b_dyn<NG, WL, HI>(ring(g), WL * AHEAD + ((has_dma && k >= KT - 1 - AHEAD) ? NG : 0));
      int rows[WM], kov[WM];
#pragma unroll
      for (int a = 0; a < WM; ++a) {
        rows[a] = arow1[a] + (G2 ? p.poff9[k] : p.poff[k]);
        // 2-D mode: tap k = (dy, dx) reads patch pixel (li + dy, lj + dx): ((lj + dx) >> 1) = (lj >> 1) + (((lj & 1) + dx) >> 1)
        int k0 = kq0[a];
        if constexpr (G2) asm volatile("" : "+v"(k0));  // recompute per tap: hoisting 18 keys out of the chunk loop spills
        kov[a] = ((k0 + hW1 * (k / 3) + (k % 3 == 2 ? 1 : 0) + (k % 3 == 1 ? kpar[a] : 0)) & 7) << 4;
      }
      mma(ring(g), lds + (c % NBUF) * PBYTES, CROW, rows, -1, kov, G2);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (c == 0) VFX_TS(13);  // (timing builds: the taps of chunk 0 / the next chunk's patch landed / its transform)
    drain();
    if (c == 0) VFX_TS(14);
    if (c + 1 < NCH) transform_patch(((c + 1) % NBUF) * PBYTES, c + 1);
    if (c == 0) VFX_TS(15);
  }

  VFX_TS(5);  // conv1 done
  __syncthreads();  // every wave is done reading the patch buffers that h overlays
  VFX_TS(6);
  }  // !SC2

  // ---- phase 2: h = LeakyReLU(conv1 + b1) in operand form, zero outside the sequence ---------------------------
  // Lane (l31, lh) of M block a holds h pixel m = (wm*WM + a)*32 + l31 and, in registers 4j .. 4j+3, channels
  // wn*32 + 8j + 4lh .. +3: their 4 hi bf16 are half `lh` of piece j of the pixel's chunk row, the 4 lo of piece j+4.
  unsigned f16_sat = 0;  // 16-bit mode: a value of h left the fp16 range and was clamped
#pragma unroll
  for (int a = 0; a < WM; ++a) {
    const int m = (wm * WM + a) * 32 + l31;
    char* rowp = lds + H_OFF + m * HROW + (NCH > 1 ? (wn ^ (m & 1)) : wn) * CROW + 8 * lh;  // chunk parity swap: see mma()
    const int key = (m >> 1) & 7;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 u;
      f32x4 hsc = {1.f, 1.f, 1.f, 1.f};  // 2-D mode: bn2 of the run's 4 channels (b1v carries the shift)
      if constexpr (G2) {
        hsc = *(const VFX_GLOBAL f32x4*)(p.sc2 + wn * 32 + 8 * j + 4 * lh);
        b1v[j] = *(const VFX_GLOBAL f32x4*)(p.sh2 + wn * 32 + 8 * j + 4 * lh);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = acc[a][4 * j + e] * hsc[e] + b1v[j][e];
        u[e] = hval[a] ? fmaxf(t, t * slope) : 0.f;
        acc[a][4 * j + e] = 0.f;
      }
      if constexpr (HI) {
        *reinterpret_cast<uint2*>(rowp + ((j ^ key) << 4)) = make_uint2(pack_f16x2(u[0], u[1], f16_sat), pack_f16x2(u[2], u[3], f16_sat));
        continue;
      }
      const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{u[0], u[1]}, bf16x2));
      const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{u[2], u[3]}, bf16x2));
      const f32x2 r01 = {u[0] - __builtin_bit_cast(float, h01 << 16), u[1] - __builtin_bit_cast(float, h01 & 0xffff0000u)};
      const f32x2 r23 = {u[2] - __builtin_bit_cast(float, h23 << 16), u[3] - __builtin_bit_cast(float, h23 & 0xffff0000u)};
      const unsigned l01 = __builtin_bit_cast(unsigned, __builtin_convertvector(r01, bf16x2));
      const unsigned l23 = __builtin_bit_cast(unsigned, __builtin_convertvector(r23, bf16x2));
      *reinterpret_cast<uint2*>(rowp + ((j ^ key) << 4)) = make_uint2(h01, h23);
      *reinterpret_cast<uint2*>(rowp + (((j + 4) ^ key) << 4)) = make_uint2(l01, l23);
    }
  }
  if constexpr (HI) report_f16_saturation(f16_sat_bits_bad(f16_sat), p.flags);
  VFX_TS(7);  // h written
  __syncthreads();  // h is complete
  VFX_TS(8);
  }  // !IN1

  // ---- phase 3: conv2 from the resident h ------------------------------------------------------------------
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const int g = NT1 + KT * c + k;
      fetch(g + AHEAD);
      if (g >= NT1 + AHEAD) wait_b_dyn<NG, WL, HI>(ring(g), WL * AHEAD);  // the first AHEAD taps landed with the last drain
      int rows[WM];
#pragma unroll
      for (int a = 0; a < WM; ++a) {
        const int r = arow2[a] + (G2 ? p.hoff9[k] : k - 1);
        rows[a] = r < 0 ? 0 : (r > MT - 1 ? MT - 1 : r);  // clamped rows only feed outputs that are masked anyway
      }
      mma(ring(g), lds + H_OFF, HROW, rows, c, rows, false);
      __builtin_amdgcn_sched_barrier(0);
    }
  drain();
  VFX_TS(9);  // conv2 done
  __syncthreads();  // every wave is done with h and the patch buffers
  VFX_TS(10);

  // ---- phase 3b (SC2): + shortcut(cat(x, x2)), a 1x1 convolution of the RAW sources -- one more 32-channel K segment per source ----
  if constexpr (SC2) {
    constexpr int NG2 = MT / RG;  // DMA instructions per wave: the tile's MT h-grid pixels, one 128-byte row each
    unsigned voff2[NG2], ok2 = 0;
#pragma unroll
    for (int q = 0; q < NG2; ++q) {  // row m = lr + RG q = h pixel (m / 16, m % 16) = image pixel (i0 - 1 + li, j0 - 1 + lj)
      const int m = lr + RG * q, li = m >> 4, lj = m & 15;
      const int r = i0 - 1 + li, c = j0 - 1 + lj;
      voff2[q] = (unsigned)((img * Hh + r) * Ww + c) * (unsigned)(C * 4);
      ok2 |= (((unsigned)r < (unsigned)Hh) & ((unsigned)c < (unsigned)Ww)) ? (1u << q) : 0u;
    }
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(sidx == 0 ? p.x : p.x2), 0, (int)(unsigned)((int64_t)p.B * Hh * Ww * C * 4), 0x00020000);
#pragma unroll
      for (int q = 0; q < NG2; ++q) {
        const unsigned o = (ok2 & (1u << q)) ? voff2[q] + 16u * cg : 0xfffffff0u;
        VFX_LDS void* l = (VFX_LDS void*)(lds + (RG * q + 8 * wave_u) * CROW);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, l, 16, (int)o, 0, 0, 0);
      }
      drain();  // this wave's rows are its own threads' rows
      {
        char* row0 = lds + lr * CROW;
        f32x4 raw[NG2];
#pragma unroll
        for (int q = 0; q < NG2; ++q) raw[q] = *reinterpret_cast<const f32x4*>(row0 + RG * q * CROW + 16 * cg);
#pragma unroll
        for (int q = 0; q < NG2; ++q) {  // hi/lo split of the raw values, row-linear swizzle key (rows lr + 32 q share it)
          const f32x4 v = raw[q];
          const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2));
          const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2], v[3]}, bf16x2));
          const f32x2 r01 = {v[0] - __builtin_bit_cast(float, h01 << 16), v[1] - __builtin_bit_cast(float, h01 & 0xffff0000u)};
          const f32x2 r23 = {v[2] - __builtin_bit_cast(float, h23 << 16), v[3] - __builtin_bit_cast(float, h23 & 0xffff0000u)};
          const unsigned l01 = __builtin_bit_cast(unsigned, __builtin_convertvector(r01, bf16x2));
          const unsigned l23 = __builtin_bit_cast(unsigned, __builtin_convertvector(r23, bf16x2));
          char* rowp = row0 + RG * q * CROW;
          const int half = 8 * (cg & 1);
          *reinterpret_cast<uint2*>(rowp + (((cg >> 1) ^ key_l) << 4) + half) = make_uint2(h01, h23);
          *reinterpret_cast<uint2*>(rowp + ((((cg >> 1) + 4) ^ key_l) << 4) + half) = make_uint2(l01, l23);
        }
      }
      __syncthreads();  // the source's centre rows are in operand form
      mma(ring(NT1 + KT + sidx), lds, CROW, arow2, -1, arow2, false);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();  // every wave is done with them (the next source / the staged output goes over them)
    }
  }

  // ---- phase 4: y = conv2 + b2 + x --------------------------------------------------------------------------
  int* otab = reinterpret_cast<int*>(lds + OTAB_OFF);
  if (tid < MT) {
    const int li = (int)(((unsigned)tid * inv_w1) >> 20), lj = tid - li * W1;
    if constexpr (G2) {  // outputs = interior of the h grid, inside the image
      const int r = i0 - 1 + li, c = j0 - 1 + lj;
      const bool ok = (li >= 1) & (li <= TH - 2) & (lj >= 1) & (lj <= W1 - 2) & (r < Hh) & (c < Ww);
      otab[tid] = ok ? (img * Hh + r) * Ww + c : -1;
    } else {
      const int pos = base_h + li * rowstride + lj;
      const bool ok = (li < TH) & (lj >= 1) & (lj <= W1 - 2) & ((unsigned)pos < (unsigned)Tb) &
                      (!p.fold | (j0 + lj - 1 < d));
      otab[tid] = ok ? img * T + pos : -1;
    }
  }
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (wm * WM + a) * 32 + l31;
      *reinterpret_cast<f32x4*>(smem + row * LDO + wn * 32 + 8 * j + 4 * lh) =
          f32x4{acc[a][4 * j], acc[a][4 * j + 1], acc[a][4 * j + 2], acc[a][4 * j + 3]};
    }
  __syncthreads();
  VFX_TS(11);  // staged
  if constexpr (KEEPRES) {
    // + x from the registers: patch pixel pr is the input sample of h pixel m = pr - d (1-D) or, folded, of the pixel one
    // patch row up -- every staged (row, 4 channels) is touched by exactly one thread
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      const int pr = lr + RG * q;
      int m;
      if (p.fold) {
        const int pi = (int)(((unsigned)pr * inv_pw) >> 20), pj = pr - pi * PW;
        m = (pi >= 1 && pi <= TH) ? (pi - 1) * W1 + pj : -1;
      } else {
        m = pr - d;
      }
      if (pr < P && m >= 0 && m < MT) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          f32x4* s4 = reinterpret_cast<f32x4*>(smem + m * LDO + c * 32 + 4 * cg);
          *s4 = *s4 + keep[c][q];
        }
      }
    }
    __syncthreads();
  }
  {
    constexpr int V = C / 4, RPP = NTHR / V, NPASS = MT / RPP;
    const int c4 = tid % V, r0 = tid / V;
    const f32x4 bv = (IN1 || SC2) ? *(const VFX_GLOBAL f32x4*)(p.bsc + 4 * c4)
                         : (G2 ? f32x4{0.f, 0.f, 0.f, 0.f} : *(const VFX_GLOBAL f32x4*)(p.b2 + 4 * c4));
    f32x4 wsv = {0.f, 0.f, 0.f, 0.f};  // entry block: the 1x1 shortcut of the single input channel
    if constexpr (IN1) wsv = *(const VFX_GLOBAL f32x4*)(p.wsc + 4 * c4);
    int opix[NPASS];
    f32x4 val[NPASS], res[NPASS];
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
      opix[q] = otab[r0 + q * RPP];
      val[q] = *reinterpret_cast<const f32x4*>(smem + (r0 + q * RPP) * LDO + 4 * c4) + bv;
    }
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
      if constexpr (IN1) {
        const float xv = *(const VFX_GLOBAL float*)(p.x + (opix[q] < 0 ? 0 : opix[q]));
        res[q] = f32x4{xv * wsv[0], xv * wsv[1], xv * wsv[2], xv * wsv[3]};
      } else if constexpr (SC2) {
        res[q] = f32x4{0.f, 0.f, 0.f, 0.f};  // the shortcut is in the accumulators
      } else {
        res[q] = KEEPRES ? f32x4{0.f, 0.f, 0.f, 0.f}  // already added in LDS
                         : *(const VFX_GLOBAL f32x4*)(p.x + (int64_t)(opix[q] < 0 ? 0 : opix[q]) * C + 4 * c4);
      }
    }
#pragma unroll
    for (int q = 0; q < NPASS; ++q) val[q] += res[q];
#pragma unroll
    for (int q = 0; q < NPASS; ++q)
      if (opix[q] >= 0) *(VFX_GLOBAL f32x4*)(p.y + (int64_t)opix[q] * C + 4 * c4) = val[q];
    if constexpr (HI && !G2) {
      // 16-bit mode, last layer of a stack: also the activated fp16 form for the upsampler that follows
      // (ya = fp16(LeakyReLU(y, act_slope)), 2 bytes per element): it then reads half the bytes and does no arithmetic on them
      if (p.ya) {
        const float aslope = p.act_slope;
        const bool even = (tid & 1) == 0;
        unsigned f16_sat = 0;
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
            *(VFX_GLOBAL f32x4*)(p.ya + (int64_t)opix[q] * (C / 2) + 2 * c4) = __builtin_bit_cast(f32x4, w);
        }
        report_f16_saturation(f16_sat_bits_bad(f16_sat), p.flags);
      }
    }
  }
  VFX_TS(12);  // stores issued
  if constexpr (G2) VFX_TS_FLUSH(p.timing, tile, wave_u, NW);
}

static size_t resblock_lds_bytes(int C, int MT = CBM) {
  const size_t h_end = (size_t)MT * C * 4;                       // h overlays the patch buffers
  const size_t epi_end = (size_t)MT * (C + 4) * 4 + MT * 4;
  const size_t patches = (size_t)(C == 32 ? 1 : (C == 128 ? 3 : 2)) * (MT + MT / 2) * CROW;  // one chunk: one buffer; four chunks: three
  return std::max(std::max(h_end, patches), epi_end);
}

template <int C, int NW, bool HI, bool G2 = false, int MT = 128, bool IN1 = false, bool SC2 = false>
static void launch_rb(int grid, hipStream_t stream, const ResBlockParams* dparams) {
#ifdef VFX_RB_EXTRA_LDS  // measurement builds: a larger LDS allocation than the kernel needs (occupancy experiments)
  const size_t lds = resblock_lds_bytes(C, MT) + VFX_RB_EXTRA_LDS;
#else
  const size_t lds = resblock_lds_bytes(C, MT);
#endif
  static uint64_t attr_devices = 0;  // one static per instantiation
  if (first_use_on_current_device(attr_devices)) {
    VFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_resblock<C, NW, HI, G2, MT, IN1, SC2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
  }
  hipLaunchKernelGGL((k_resblock<C, NW, HI, G2, MT, IN1, SC2>), dim3(grid), dim3(NW * 64), lds, stream, dparams);
}

bool resblock_supported(int C) { return C == 64 || C == 128; }

// Waves per block of the kernel that runs this (planned) layer: the second template argument in the kernel tables
int resblock_block_waves(const ResBlockParams& hp) {
  if (hp.rw) return hp.tile_m / 32;
  if (hp.asrc || hp.r128) return 4;
  if (hp.geo2d) return (hp.C == 32 && hp.tile_m != 256) ? 2 : 4;
  return hp.C >= 128 ? 8 : 4;
}
bool block2d_supported(int C) { return C == 32 || C == 64; }

// Tile geometry of a fused 2-D ConvBlockRes (B, H, W, C must be set): h grid TH x W1 = 128 pixels in the orientation that
// wastes fewer pixels on the image borders, outputs = its interior, x patch = (TH + 2) x (W1 + 2).
void plan_block2d(ResBlockParams& p) {
  VFX_CHECK(block2d_supported(p.C), "block2d: C=%d is not supported", p.C);
  // the tile that computes the fewest h positions per image: 128 positions as 8 x 16 or 16 x 8 (6 x 14 outputs); at C = 32 also
  // 256 positions as 16 x 16 (14 x 14 outputs: k_resblock<32, 4, .., MT = 256>; VFX_TUNE_SMALL_2D_TILES keeps the 128-position ones)
  double best = -1.0;
  p.tile_m = 0;
  struct Cand { int TH, W1; };
  std::vector<Cand> cands = {{8, 16}, {16, 8}};
  if (p.C == 32 && !(p.tuning & VFX_TUNE_SMALL_2D_TILES)) cands.push_back({16, 16});
  // 14 x 18 (12 x 16 outputs; round 6): in the persistent kernel only (block2d32.hip) -- 128 mel bins are 8 tile columns instead of 10
  if (p.C == 32 && !(p.tuning & (VFX_TUNE_SMALL_2D_TILES | VFX_TUNE_OLD_BLOCK2D)) && !p.in1 && !p.two_src && !p.hionly && p.W <= 4096 &&
      p.H <= 65536) {
    const double tpi = (double)((p.H + 11) / 12) * ((p.W + 15) / 16);
    if ((double)p.B * tpi * tpi < 4294967296.0) cands.push_back({14, 18});  // (block2d32_ok: the reciprocal tile split must be exact)
  }
  if (p.in1 || p.two_src) {  // the entry block (Cin = 1) and the two-source block exist on 16 x 16 tiles only
    VFX_CHECK(p.C == 32 && !(p.tuning & VFX_TUNE_SMALL_2D_TILES), "block2d: the entry / two-source block runs C = 32 on 16 x 16 tiles");
    cands = {{16, 16}};
  }
  for (const Cand& c : cands) {
    const int oh = c.TH - 2, ow = c.W1 - 2;
    const int slots = c.TH * c.W1 > CBM ? 256 : CBM;  // (a 14 x 18 tile costs what a 16 x 16 tile costs)
    const double positions = (double)((p.H + oh - 1) / oh) * ((p.W + ow - 1) / ow) * slots;  // h positions computed
    const double util = (double)p.H * p.W / positions;
    if (util > best) {
      best = util;
      p.W1 = c.W1;
      p.TH = c.TH;
      p.tile_m = slots == 256 ? 256 : 0;
    }
  }
#ifdef VFX_TIMING
  p.timing = getenv("VFX_TIMING_PTR") ? reinterpret_cast<unsigned long long*>(strtoull(getenv("VFX_TIMING_PTR"), nullptr, 0)) : nullptr;
#endif
  p.geo2d = 1;
  p.fold = 0;
  p.dil = 1;
  p.T = p.H * p.W;
  p.TWo = p.W1 - 2;
  p.PW = p.W1 + 2;
  p.P = (p.TH + 2) * p.PW;
  p.tiles_h = (p.H + p.TH - 3) / (p.TH - 2);
  p.tiles_w = (p.W + p.TWo - 1) / p.TWo;
  for (int dy = 0; dy < 3; ++dy)
    for (int dx = 0; dx < 3; ++dx) {
      p.poff9[dy * 3 + dx] = dy * p.PW + dx;                 // conv1: x patch rows of the tap, relative to the h pixel's row
      p.hoff9[dy * 3 + dx] = (dy - 1) * p.W1 + (dx - 1);    // conv2: h rows
    }
  const int MT2 = p.tile_m == 256 ? 256 : CBM;
  VFX_CHECK(p.P <= MT2 + MT2 / 2 && (p.TH * p.W1 == MT2 || (p.TH == 14 && p.W1 == 18 && MT2 == 256)), "block2d: bad tile geometry");
  VFX_CHECK((int64_t)p.B * p.H * p.W * p.C * 4 < ((int64_t)1 << 32) - 4096, "block2d: tensor exceeds 4 GiB");
  VFX_CHECK(!(p.in1 || p.two_src) || p.tile_m == 256, "block2d: entry / two-source block tile");
  // tile -> (image, tile row, tile column) of the persistent kernel (block2d32.hip), cf. plan_resblock
  const int64_t tpi = (int64_t)p.tiles_w * p.tiles_h;
  p.inv_tiles_w = (((uint64_t)1 << 32) + p.tiles_w - 1) / p.tiles_w;
  p.inv_tiles_per_img = (((uint64_t)1 << 32) + tpi - 1) / tpi;
  p.recip_ok = (double)p.B * (double)tpi * (double)tpi < 4294967296.0 ? 1 : 0;
}

// Fills the tile geometry of a fused ResStack layer (B, T, C, dil must be set).
void plan_resblock(ResBlockParams& p) {
  VFX_CHECK(p.asrc ? resblock_w64_supported(p.C) : resblock_supported(p.C), "resblock: C=%d is not supported", p.C);
  const int d = p.dil;
#ifdef VFX_TIMING
  p.timing = getenv("VFX_TIMING_PTR") ? reinterpret_cast<unsigned long long*>(strtoull(getenv("VFX_TIMING_PTR"), nullptr, 0)) : nullptr;
#endif
  // 16-bit mode, C = 64: the persistent register-weights kernel (resblock_rw.hip) with its own tile size
  p.rw = (!p.asrc && p.hionly && p.C == 64 && resblock_rw_tile(p.tuning) != 0) ? 1 : 0;
  if (p.rw) p.tile_m = resblock_rw_tile(p.tuning);
#ifdef VFX_RW_SINGLE_MT
  if (p.rw && p.dil2 == 0 && p.x16) p.tile_m = VFX_RW_SINGLE_MT;  // measurement builds: single layers on two 4-wave blocks per CU
#endif
  const int MT = p.tile_m ? p.tile_m : CBM;  // h positions per tile (resblock_rw: 128 or 256)
  // patch rows per buffer: MT + 64 (= kPatchMaxRows for MT = 128); the 4-wave form of the wide layer keeps four chunk
  // buffers in half a CU's LDS: 160 rows (resblock_w64.hip)
  VFX_CHECK(!p.asrc || (MT == 128 && p.dil2 == 0), "resblock: the wide layer runs 128-position tiles, one layer per launch");
  p.patch_rows = p.asrc ? resblock_w64_patch_rows() : 0;
  p.r128 = (!p.asrc && !p.rw && p.hionly && p.C == 128 && MT == 128 &&
            (p.dil2 == 0 || resblock_r128_pair_ok(p.C, d, p.dil2, p.tuning))) ? 1 : 0;
  if (p.r128) p.patch_rows = resblock_r128_patch_rows();
  // the fp16 trunk exists in the kernels of the 16-bit mode only (resblock_rw / resblock_r128 / resblock_w64)
  VFX_CHECK(!p.x16 || p.rw || p.r128 || p.asrc, "resblock: no kernel runs this layer (C = %d) on an fp16 trunk", p.C);
  const int PR = p.patch_rows ? p.patch_rows : MT + 64;
  VFX_CHECK(MT == 128 || (MT == 256 && p.rw), "resblock: tile of %d positions", MT);
  p.tile_m = MT;
  if (p.dil2 > 0) {
    // layer pair: both layers over the MT-index space of the tile, MT - 4 - 2 dil2 outputs per tile (resblock_rw.hip)
    VFX_CHECK((p.rw && resblock_rw_pair_ok(p.C, d, p.dil2, p.tuning)) || (p.r128 && MT + 2 * d <= PR && MT + 2 * p.dil2 <= PR),
              "resblock: layers of dilation %d, %d cannot run as a pair", d, p.dil2);
    p.fold = 0;
    p.TH = 1;
    p.W1 = MT;
    p.TWo = MT - 4 - 2 * p.dil2;
    p.tiles_h = 1;
    p.tiles_w = (p.T + p.TWo - 1) / p.TWo;
    p.PW = MT + 2 * d;
    p.P = p.PW;
    for (int k = 0; k < 3; ++k) p.poff[k] = k * d;
  } else if (MT + 2 * d <= PR) {
    p.fold = 0;
    p.TH = 1;
    p.W1 = MT;
    p.TWo = MT - 2;
    p.tiles_h = 1;
    p.tiles_w = (p.T + p.TWo - 1) / p.TWo;
    p.PW = MT + 2 * d;
    p.P = p.PW;
    for (int k = 0; k < 3; ++k) p.poff[k] = k * d;
  } else {
    // rows of d samples; h tile TH x (TW + 2) <= MT pixels, x patch (TH + 2) x (TW + 2) <= PR
    p.fold = 1;
    // tile width: the widest (<= 16) that cuts a row of d samples into equal parts -- d = 81 as 6 x 16 wastes 15 of 96 columns
    // (the d = 81 layers ran 15 % longer than the other folded ones), as 6 x 14 it wastes 3; the h tile is 16 x 16 instead of 14 x 18
    // (160-row patches: 14-wide tiles at most -- the h tile is then 8 x 16 = all 128 positions; 16-wide tiles would be 6 x 18)
    const int rows = (p.T + d - 1) / d;
    auto geometry = [&](int twmax, int pr, ResBlockParams& q) -> int64_t {  // fills q, returns the tiles per sequence
      const int nparts = (d + twmax - 1) / twmax;
      const int TW = d >= twmax ? (d + nparts - 1) / nparts : d;
      q.W1 = TW + 2;
      const int th_max = std::min(MT / q.W1, pr / q.W1 - 2);
      if (th_max < 1) return (int64_t)1 << 60;
      q.tiles_h = (rows + th_max - 1) / th_max;
      q.TH = (rows + q.tiles_h - 1) / q.tiles_h;  // equal parts here too: 23 rows are 4 x 6, not 4 x 7
      q.TWo = TW;
      q.tiles_w = (d + TW - 1) / TW;
      q.PW = q.W1;
      q.P = (q.TH + 2) * q.W1;
      for (int k = 0; k < 3; ++k) q.poff[k] = k * q.W1;
      return (int64_t)q.tiles_h * q.tiles_w;
    };
    int64_t tiles = geometry(p.patch_rows ? 14 : 16, PR, p);
    if (p.rw && p.x16 && MT == 256) {
      // persistent C = 64 kernel on the fp16 trunk: tiles up to 62 wide on a (MT + 128)-row patch (resblock_rw.hip, HALO = 128)
      // where they need fewer tiles -- d = 243 .. 2187: 4 x 63 h tiles with 244 outputs instead of 14 x 18 with 224
      ResBlockParams q = p;
      const int64_t wide = geometry(62, MT + 128, q);
      if (wide < tiles) {
        q.patch_rows = MT + 128;
        p = q;
        tiles = wide;
      }
    }
  }
  const int PRc = p.patch_rows ? p.patch_rows : MT + 64;  // (the wide folded tiles of resblock_rw enlarge the patch)
  VFX_CHECK(p.P <= PRc && p.TH * p.W1 <= MT && p.TH >= 1, "resblock: bad tile geometry (dil=%d)", d);
  p.inv_pw = ((1u << 20) + p.PW - 1) / p.PW;
  p.inv_w1 = ((1u << 20) + p.W1 - 1) / p.W1;
  const int64_t tpi = (int64_t)p.tiles_w * p.tiles_h;
  p.inv_tiles_w = (((uint64_t)1 << 32) + p.tiles_w - 1) / p.tiles_w;  // = 2^32 for one tile: does not fit 32 bits
  p.inv_tiles_per_img = (((uint64_t)1 << 32) + tpi - 1) / tpi;
  // exact while n * d < 2^32: n = tile index < B * tpi, d = tpi
  VFX_CHECK(!p.patch_rows || (double)p.B * (double)tpi * (double)tpi < 4294967296.0, "resblock: too many tiles for the reciprocal tile split");
  VFX_CHECK((int64_t)p.B * p.T * p.C * 4 < ((int64_t)1 << 32) - 4096, "resblock: tensor exceeds 4 GiB");
}

void launch_resblock(const ResBlockParams& hp, const ResBlockParams* dparams, hipStream_t stream) {
  if (hp.asrc) {
    launch_resblock_w64(hp, dparams, stream);
    return;
  }
  if (hp.rw) {
    launch_resblock_rw(hp, dparams, stream);
    return;
  }
  if (hp.r128) {
    launch_resblock_r128(hp, dparams, stream);
    return;
  }
  const int64_t grid = (int64_t)hp.B * hp.tiles_h * hp.tiles_w;
  VFX_CHECK(grid > 0 && grid < ((int64_t)1 << 31), "resblock: bad grid");
  if (hp.geo2d) {
    VFX_CHECK(!hp.hionly, "block2d: split-bf16 only");
    // C = 32: two waves of 64 pixels, not four of 32 -- every wave of a block fetches ALL the weight fragments, so fewer,
    // larger waves halve that traffic (measured -12 %; the timing-only build without weight refreshes ran this block 25 % faster)
    if (block2d32_ok(hp)) {
      launch_block2d32(hp, dparams, stream);
      return;
    }
    if (hp.in1) launch_rb<32, 4, false, true, 256, true>((int)grid, stream, dparams);
    else if (hp.two_src) launch_rb<32, 4, false, true, 256, false, true>((int)grid, stream, dparams);
    else if (hp.C == 32 && hp.tile_m == 256) launch_rb<32, 4, false, true, 256>((int)grid, stream, dparams);
    else if (hp.C == 32) launch_rb<32, 2, false, true>((int)grid, stream, dparams);
    else launch_rb<64, 4, false, true>((int)grid, stream, dparams);
    VFX_HIP(hipGetLastError());
    return;
  }
  if (hp.C == 64) {
    if (hp.hionly) launch_rb<64, 4, true>((int)grid, stream, dparams);
    else launch_rb<64, 4, false>((int)grid, stream, dparams);
  } else {
    VFX_CHECK(!hp.hionly, "resblock: the C = 128 layers of the 16-bit mode run on resblock_r128");
    launch_rb<128, 8, false>((int)grid, stream, dparams);
  }
  VFX_HIP(hipGetLastError());
}

double resblock_flops(const ResBlockParams& hp) {
  if (hp.in1) return 2.0 * (double)hp.B * hp.T * hp.C * 9.0 * (1.0 + hp.C);  // conv1 has one input channel
  if (hp.two_src) return 2.0 * (double)hp.B * hp.T * hp.C * (9.0 * 3.0 * hp.C + 2.0 * hp.C);  // conv1 over 2 C, conv2, the 1x1 shortcut
  return (hp.dil2 > 0 ? 2.0 : 1.0) * 2.0 * 2.0 * (double)hp.B * hp.T * hp.C * ((hp.geo2d ? 9.0 : 3.0) * hp.C);  // pairs: two layers
}

}  // namespace vfx
