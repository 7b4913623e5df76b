// block2d32.hip -- the fused 2-D ConvBlockRes of ResUNet level 1 (models/components/modules.py:223-271, Cin == Cout == 32, identity
// shortcut) as a PERSISTENT kernel with a lean tile loop (round 6):
//
//     y = x + conv2(lrelu(bn2(conv1(lrelu(bn1(x))))))        3x3 convolutions, split-bf16 operands
//
// Same arithmetic as k_resblock<32, 4, false, true, 256> (resblock.hip) -- the same products summed in the same order, the residual added
// last: BIT-IDENTICAL (tests/test_gpu_kernels.py::test_block2d32_equals_the_one_tile_per_block_kernel) -- on an h grid of 16 x 16 pixels
// (14 x 14 outputs, x patch 18 x 18) or 14 x 18 (12 x 16 outputs, patch 16 x 20), four waves of 64 h pixels x all 32 channels, three blocks
// per CU.  What is different is what a tile costs in INSTRUCTIONS.  Phase stamps + instruction counts of the old kernel
// (profiles/r06_c20_block2d_phase_stamps.txt): 1 368 VALU, 297 SALU, 212 LDS and 118 VMEM instructions around 216 MFMAs per tile and
// wave, block lifetime 36.3 k cycles.  Here:
//   * blocks are persistent (768 of them walk the tiles): everything that does not depend on the tile -- which patch pixels a thread
//     stages, where they land in LDS, which LDS rows a lane's fragments come from, where its outputs go -- is computed once per block;
//     per tile only two buffer descriptors (scalar arithmetic) change;
//   * LDS rows are PADDED (144 bytes per pixel: 64 of hi halves, 64 of lo halves, 16 unused; patch rows of 2 816 / 3 104 bytes), not
//     swizzled: pixels whose addresses differ by an odd multiple of 16 bytes (mod 256) never share a bank quad, so the ds_read_b128 lane
//     groups of gfx950 ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}: consecutive pixels of two tile rows) are conflict-free without any
//     XOR -- and every tap, every K step and the hi / lo halves are IMMEDIATE offsets of ONE address register per 32-pixel block.  The
//     two convolution loops contain no VALU instruction at all;
//   * the patch goes global -> registers -> (bn1, LeakyReLU, hi/lo split) -> LDS instead of LDS-DMA + in-place rewrite: no raw landing
//     zone, no second LDS pass, and the next tile's pixels are requested before the barrier that ends the current tile;
//   * interior tiles (78 % at level 1) run without any validity mask; border tiles take a second, masked copy of the two VALU phases;
//   * conv1 starts from the MFMA's zero constant instead of 32 cleared registers; the residual is requested late in conv2 (L2-hot) and
//     added in the epilogue, where the tile is staged through LDS and leaves in full 128-byte lines (stored straight from the MFMA
//     layout every store instruction touches 32 lines: 0.12 ms of a 0.33 ms launch);
//   * weights: ordinary global loads into a ring of three register groups (the compiler counts vmcnt: there is no LDS-DMA beside them).
// Measured: profiles/r06_c40_block2d32_ablations.txt; DESIGN.md section 5.
#include "conv_common.h"
#include "vfx_internal.h"

#include <type_traits>


namespace vfx {

namespace b2d {
constexpr int PIX = 144;            // LDS bytes per pixel: [32 hi bf16 | 32 lo bf16 | 16 pad]
// Geometry of a tile: an h grid of TH x W1 <= 256 pixels (16 x 16, or 14 x 18: 12 x 16 outputs divide the 128 mel bins of level 1 without a
// remainder -- 680 tiles per image instead of 730), outputs = its interior, patch (TH + 2) x (W1 + 2).
template <int TH, int W1>
struct Geo {
  static constexpr int OH = TH - 2, OW = W1 - 2, PH = TH + 2, PW = W1 + 2, NPIX = PH * PW;
  static constexpr int NQ = (NPIX + 31) / 32;          // patch pixels per thread: pixel lr + 32 q
  static constexpr int LASTQ = NPIX - 32 * (NQ - 1);   // pixels of the last q (threads lr < LASTQ)
  // LDS bytes per patch row: lanes of a ds_read_b128 group are consecutive h pixels ml = W1 li + lj of one or two grid rows; their
  // 16-byte units li * (PITCH / 16) + 9 lj = 9 ml + li (PITCH / 16 - 9 W1) are distinct mod 16 iff PITCH / 16 = 9 W1 (mod 16)
  static constexpr int pitch_units() {
    int u = 9 * PW;
    while (u % 16 != (9 * W1) % 16) ++u;
    return u;
  }
  static constexpr int PITCH = 16 * pitch_units();     // 2 816 (16 x 16), 3 104 (14 x 18)
  static constexpr int PATCH_BYTES = PH * PITCH;       // 50 688, 49 664
  static constexpr int HOFF = (W1 + 1) * PIX;          // the h grid (256 slots of PIX bytes) overlays the patch; W1 + 1 pixels of slack either
                                                       // side: conv2's taps of the grid's border pixels read bytes that only feed outputs nobody stores
  static constexpr int BN_OFF = PATCH_BYTES;           // bn1 scale, bn1 shift, bn2 scale, bn2 shift: 32 floats each
  static constexpr int LDS_BYTES = PATCH_BYTES + 512;
  // the staged tile: outputs only, rows of 16 slots (OW <= 16), NE x 32 slots + one dummy slot for the lanes that hold no output
  static constexpr int NE = (OH * 16 + 31) / 32;       // 7, 6: output pieces per thread
  static_assert(TH * W1 <= 256 && OW <= 16 && HOFF + (256 + W1 + 1) * PIX <= PATCH_BYTES && (NE * 32 + 1) * PIX <= PATCH_BYTES &&
                    2 * (NE - 1) + 1 <= OH - 1 && LDS_BYTES <= 53248, "tile geometry");
};
constexpr unsigned kOob = 0x80000000u;    // a byte offset past every descriptor's num_records: loads return zeros, stores are dropped
constexpr int kNumRecords = 0x40000000;
constexpr int RING = 3, AHEAD = RING - 1;
constexpr int kResStep = 16;        // conv2 requests the residual (L2-hot: the patch read the same lines) in front of this K step of 18:
                                    // late, when two of the three weight groups are dead -- any earlier and the kernel spills
}  // namespace b2d

__device__ __forceinline__ void split_bf16x4(const f32x4 v, uint2& hi, uint2& lo) {
  const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2));
  const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2], v[3]}, bf16x2));
  const f32x2 r01 = {v[0] - __builtin_bit_cast(float, h01 << 16), v[1] - __builtin_bit_cast(float, h01 & 0xffff0000u)};
  const f32x2 r23 = {v[2] - __builtin_bit_cast(float, h23 << 16), v[3] - __builtin_bit_cast(float, h23 & 0xffff0000u)};
  hi = make_uint2(h01, h23);
  lo = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(r01, bf16x2)),
                  __builtin_bit_cast(unsigned, __builtin_convertvector(r23, bf16x2)));
}

#ifndef VFX_B2D_LB  // measurement builds (scripts/build_variant.sh): blocks per CU the register budget is cut for / the grid is sized for
#define VFX_B2D_LB 3
#endif
#ifndef VFX_B2D_PER_CU
#define VFX_B2D_PER_CU 3
#endif
#define B2D_BARRIER() do { if (!(VFX_B2D_ABL & 64)) __syncthreads(); } while (0)
#ifndef VFX_B2D_ABL  // timing-only ablations (wrong results): 1 no patch loads, 2 no MFMAs, 4 no stores, 8 no fragment reads, 16 no residual
#define VFX_B2D_ABL 0  // loads, 32 no LDS writes of the two VALU phases, 64 no barriers
#endif
template <int TH, int W1>
__global__ __launch_bounds__(256, VFX_B2D_LB) void k_block2d32(const ResBlockParams* __restrict__ pp, int ntiles) {
  using namespace b2d;
  using G = Geo<TH, W1>;
  constexpr int PW = G::PW, NPIX = G::NPIX, NQ = G::NQ, PITCH = G::PITCH, HOFF = G::HOFF, BN_OFF = G::BN_OFF, NE = G::NE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* const lds = reinterpret_cast<char*>(smem);
  const ResBlockParams& p = *pp;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int lr = tid >> 3, cg = tid & 7;
  const int Hh = p.H, Ww = p.W;
  const float slope = p.slope;

  // ---- what never changes from tile to tile ------------------------------------------------------------------------------------------
  // patch pixel lr + 32 q = (pi, pj) of the 18 x 18 window: its byte offset from the window's first pixel in x, and where its
  // 4 channels 4 cg .. 4 cg + 3 go in LDS (hi halves at + 0, lo halves at + 64)
  // Kept compact (22 table registers would not fit beside the accumulators, the weight ring and two fragment buffers): the window row
  // of pixel lr + 32 q is pi = (32 q) / 18 + c_q with c_q = (lr + (32 q) % 18) / 18 in {0, 1, 2} -- two bits per q in `cbits` -- and
  //   x offset  = (lr + 32 q) * 128 + 16 cg + pi * 128 (W - 18)  = vbase + [q * 4096 + (32 q / 18) * wd] + c_q * wd    ([..]: scalar)
  //   LDS offset = (lr + 32 q) * 144 + 8 cg + pi * (PITCH - 18 PIX) = wbase + [constant] + c_q * 224
  unsigned cbits = 0;
#pragma unroll
  for (int q = 0; q < NQ; ++q) cbits |= (unsigned)((lr + (32 * q) % PW) / PW) << (2 * q);
  const unsigned vbase = (unsigned)lr * 128u + 16u * (unsigned)cg;
  const int wbase = lr * PIX + 8 * cg;
  const unsigned wd = (unsigned)__builtin_amdgcn_readfirstlane(128 * (Ww - PW));
  unsigned cb = cbits;  // (re-defined opaquely per tile: otherwise the compiler hoists the eleven offsets out of the tile loop and spills them)
  auto voff = [&](int q) __attribute__((always_inline)) -> unsigned {   // (without the scalar part: xsoff())
    const unsigned v = vbase + ((cb >> (2 * q)) & 3u) * wd;
    return (q == NQ - 1 && lr >= G::LASTQ) ? kOob : v;              // (16 x 16: 324 = 10 x 32 + 4 pixels)
  };
  auto xsoff = [&](int q) __attribute__((always_inline)) -> int {
    return __builtin_amdgcn_readfirstlane(q * 4096 + ((32 * q) / PW) * (int)wd);  // (an SGPR operand: without this the compiler parks the
  };                                                                              // loop-invariant sums in VGPRs and builds waterfall loops)
  auto wr = [&](int q) __attribute__((always_inline)) -> int {
    return wbase + 32 * q * PIX + ((32 * q) / PW) * (PITCH - PW * PIX) + (int)((cb >> (2 * q)) & 3u) * (PITCH - PW * PIX);
  };
  // h pixel of this lane in M block a: ml = wave * 64 + 32 a + l31 = (li, lj) of the 16 x 16 grid
  int a1[2], a2[2], hw[2], sw[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int ml = wave * 64 + 32 * a + l31;
    const bool in_grid = ml < TH * W1;              // (14 x 18: slots 252 .. 255 hold no h pixel; they compute on pixel 0 and store nothing)
    const int li = in_grid ? ml / W1 : 0, lj = in_grid ? ml - li * W1 : 0;
    a1[a] = li * PITCH + lj * PIX + 16 * lh;        // conv1: tap (dy, dx) at + dy * PITCH + dx * PIX, K step s at + 32 s, lo halves at + 64
    a2[a] = HOFF + (ml - (W1 + 1)) * PIX + 16 * lh; // conv2: tap (dy, dx) at + (W1 dy + dx) * PIX
    hw[a] = HOFF + ml * PIX + 8 * lh;               // h write: channel run j at + 16 j, lo halves at + 64
    // staged tile: output (oi, oj) = h pixel (oi + 1, oj + 1) at slot 16 oi + oj; the lanes of the grid's border at the dummy slot
    const bool out = in_grid && li >= 1 && li <= TH - 2 && lj >= 1 && lj <= W1 - 2;
    sw[a] = (out ? 16 * (li - 1) + (lj - 1) : NE * 32) * PIX + 16 * lh;
  }
  // Epilogue roles (the staged tile leaves through full 128-byte lines, as the patch came in): thread (lr, cg) adds the residual to and
  // stores piece cg of the outputs at slots lr + 32 q = (oi, oj) = ((lr >> 4) + 2 q, lr & 15), q = 0 .. NE - 1;
  // output (oi, oj) = patch pixel (oi + 2, oj + 2): byte offset e0 + q * 256 W from the patch window's first pixel
  const int elj = lr & 15, eli0 = lr >> 4;
  const unsigned e0 = elj < G::OW ? (unsigned)((eli0 + 2) * Ww + elj + 2) * 128u + 16u * (unsigned)cg : kOob;
  const int erow = __builtin_amdgcn_readfirstlane(256 * Ww);
  const int srd = lr * PIX + 16 * cg;   // staged tile: slot n at n * PIX (over h: every wave has passed the barrier behind conv2)
  if (tid < 32) {  // the folded BatchNorm affines: bn1 scale | bn1 shift | bn2 scale | bn2 shift, 32 floats each
    const float* src = tid < 8 ? p.sc1 : (tid < 16 ? p.sh1 : (tid < 24 ? p.sc2 : p.sh2));
    *reinterpret_cast<f32x4*>(lds + BN_OFF + 16 * tid) = *(const VFX_GLOBAL f32x4*)(src + 4 * (tid & 7));
  }
  // weights: one descriptor per convolution, tap g's fragment i at byte g * 4096 + i * 1024 + lane * 16 (pack_conv's fragment order)
  const __amdgpu_buffer_rsrc_t rw1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w1), 0, 9 * 4096, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w2), 0, 9 * 4096, 0x00020000);
  const int lane16 = lane * 16;

  // ---- the tiles of this block: every XCD (blockIdx % 8) walks a contiguous range, its blocks take consecutive tiles ----------------
  const int nblk = gridDim.x >> 3;       // blocks per XCD (the launch makes gridDim.x a multiple of 8)
  int t, t_end;
  {
    const int xcd = blockIdx.x & 7, q = ntiles >> 3, r = ntiles & 7;
    const int lo = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    t = lo + (blockIdx.x >> 3);
    t_end = lo + q + (xcd < r ? 1 : 0);
  }
  if (t >= t_end) return;

  // ---- weights: tap g of a tile (conv1: 0 .. 8, conv2: 9 .. 17) in ring group g % 3 (18 % 3 == 0: the same group in every tile) ------
  f32x4 Wr[RING][4];
  bool wfirst = true;  // (measurement builds only)
  auto fetch = [&](int g) __attribute__((always_inline)) {
#ifdef VFX_B2D_ABL_NOWEIGHTS  // timing-only build (wrong results): the ring is filled during the first tile and never refreshed
    if (!wfirst) return;
#endif
    const int gg = g % 18;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      Wr[g % RING][i] = __builtin_bit_cast(
          f32x4, __builtin_amdgcn_raw_buffer_load_b128(gg < 9 ? rw1 : rw2, lane16 + 1024 * i, (gg < 9 ? gg : gg - 9) * 4096, 0));
  };

  // ---- per tile: descriptors (scalar), validity masks of border tiles ---------------------------------------------------------------------
  f32x4 raw[NQ];
  unsigned okbits = 0;   // border tiles: bit q = patch pixel q lies inside the image, bit 16 + a = h pixel of M block a does
  bool border = false;
  __amdgpu_buffer_rsrc_t rx, ry;       // x and y seen from the patch window's first pixel (i0 - 2, j0 - 2)
  int ti0 = 0, tj0 = 0;                // first output row / column of the tile (border tiles: the epilogue's image bounds)
  auto begin_tile = [&](int tile) __attribute__((always_inline)) {
    cb = cbits;
    asm volatile("" : "+v"(cb));
#pragma unroll
    for (int g = 0; g < AHEAD; ++g) fetch(g);  // conv1's first taps: older than the patch requests, landed when the patch has
    const int tpi = p.tiles_w * p.tiles_h;
    const int img = div_recip(tile, p.inv_tiles_per_img);
    const int rem = tile - img * tpi;
    const int ti = div_recip(rem, p.inv_tiles_w), tj = rem - ti * p.tiles_w;
    const int i0 = ti * G::OH, j0 = tj * G::OW;
    ti0 = i0;
    tj0 = j0;
    border = (i0 < 2) | (i0 + TH > Hh) | (j0 < 2) | (j0 + W1 > Ww);
    const int64_t px = ((int64_t)img * Hh + (i0 - 2)) * Ww + (j0 - 2);  // first pixel of the patch window (may lie outside the tensor)
    rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x) + px * 32, 0, kNumRecords, 0x00020000);
    ry = __builtin_amdgcn_make_buffer_rsrc(p.y + px * 32, 0, kNumRecords, 0x00020000);
    if (border) {
      okbits = 0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int prow = lr + 32 * q;
        const int pi = prow / PW, pj = prow - pi * PW;
        const bool ok = (prow < NPIX) & ((unsigned)(i0 - 2 + pi) < (unsigned)Hh) & ((unsigned)(j0 - 2 + pj) < (unsigned)Ww);
        okbits |= ok ? (1u << q) : 0u;
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        raw[q] = __builtin_bit_cast(
            f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)((okbits & (1u << q)) ? voff(q) : kOob), xsoff(q), 0));
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int ml = wave * 64 + 32 * a + l31, li = ml / W1, lj = ml - li * W1;
        const bool in = ((unsigned)(i0 - 1 + li) < (unsigned)Hh) & ((unsigned)(j0 - 1 + lj) < (unsigned)Ww);
        okbits |= in ? (1u << (16 + a)) : 0u;
      }
    } else {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if (VFX_B2D_ABL & 1) asm volatile("" : "=v"(raw[q]));
        else raw[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)voff(q), xsoff(q), 0));
      }
    }
  };

  f32x16 acc[2];
  f32x4 res[NE];    // the residual: x at this thread's output pieces (epilogue roles), requested late in conv2
  // byte offset of output piece q (without the scalar part q * erow), out of range where nothing is stored
  auto eoff = [&](int q, bool brd) __attribute__((always_inline)) -> unsigned {
    // the image bounds, on every tile (two compares; a branch on `brd` here made hipcc 7.2 emit two-path code around every second load)
    (void)brd;
    const bool in = (ti0 + eli0 + 2 * q < Hh) & (tj0 + elj < Ww);
    return in ? e0 : kOob;
  };
  // One convolution: nine taps of the LDS image behind `base[a]`; tap (dy, dx) sits `dyb` bytes per row and PIX bytes per column further
  // on.  18 K steps (tap k, channels 16 s .. 16 s + 15) of four fragment reads and six MFMAs; the reads of step i + 1 are issued
  // before the MFMAs of step i (left alone the compiler issues every read right in front of the MFMA that needs it).
  auto conv = [&](const int (&base)[2], int dyb, int g0, auto second_tag) __attribute__((always_inline)) {
    constexpr bool SECOND = decltype(second_tag)::value;
    bf16x8 F[2][4];  // ah[0], ah[1], al[0], al[1] of a step
    auto frags = [&](int i, bf16x8 (&f)[4]) __attribute__((always_inline)) {
      const int k = i >> 1, sb = 32 * (i & 1), off = (k / 3) * dyb + (k % 3) * PIX;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        if (VFX_B2D_ABL & 8) {
          asm volatile("" : "=v"(f[a]), "=v"(f[2 + a]));
          continue;
        }
        f[a] = *reinterpret_cast<const bf16x8*>(lds + base[a] + off + sb);
        f[2 + a] = *reinterpret_cast<const bf16x8*>(lds + base[a] + off + sb + 64);
      }
    };
    frags(0, F[0]);
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      const int k = i >> 1, sidx = i & 1;
      if (sidx == 0 && g0 + k + AHEAD < 18) fetch(g0 + k + AHEAD);  // (the next tile's first taps come with its patch: begin_tile)
      if (SECOND && i == kResStep) {
#pragma unroll
        for (int q = 0; q < NE; ++q)
          if (VFX_B2D_ABL & 16) asm volatile("" : "=v"(res[q]));
          else res[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)eoff(q, border), q * erow, 0));
      }
      if (i + 1 < 18) frags(i + 1, F[(i + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8 bh = __builtin_bit_cast(bf16x8, Wr[(g0 + k) % RING][2 * sidx]);
      const bf16x8 bl = __builtin_bit_cast(bf16x8, Wr[(g0 + k) % RING][2 * sidx + 1]);
      const bf16x8(&f)[4] = F[i & 1];
      // D = W (A operand: rows = couts) x pixels (B operand): small cross terms first, the dominant hi * hi product last
      if (VFX_B2D_ABL & 2) {  // keep the operands live, no MFMA
        asm volatile("" : : "v"(bh), "v"(bl), "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]));
        if (i == 0) asm volatile("" : "=v"(acc[0]), "=v"(acc[1]));
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        if (i == 0) {
          const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, f[a], z, 0, 0, 0);  // (the MFMA's zero constant: no cleared registers)
        } else {
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, f[a], acc[a], 0, 0, 0);
        }
      }
#pragma unroll
      for (int a = 0; a < 2; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, f[2 + a], acc[a], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 2; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, f[a], acc[a], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- the two VALU phases, with (border tiles) and without validity masks -----------------------------------------------------------
  auto transform = [&](auto border_tag) __attribute__((always_inline)) {
    constexpr bool BORDER = decltype(border_tag)::value;
    const f32x4 psc = *reinterpret_cast<const f32x4*>(lds + BN_OFF + 16 * cg);
    const f32x4 psh = *reinterpret_cast<const f32x4*>(lds + BN_OFF + 128 + 16 * cg);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float u = raw[q][e] * psc[e] + psh[e];
        v[e] = fmaxf(u, u * slope);
        if (BORDER) v[e] = (okbits & (1u << q)) ? v[e] : 0.f;  // zero padding AFTER bn1 + LeakyReLU
      }
      uint2 hi, lo;
      split_bf16x4(v, hi, lo);
      if (q < NQ - 1 || lr < G::LASTQ) {  // (16 x 16, q = 10: the window's last four pixels)
        const int w = wr(q);
        if (VFX_B2D_ABL & 32) {
          asm volatile("" : : "v"(hi), "v"(lo), "v"(w));
          continue;
        }
        *reinterpret_cast<uint2*>(lds + w) = hi;
        *reinterpret_cast<uint2*>(lds + w + 64) = lo;
      }
    }
  };
  auto hwrite = [&](auto border_tag) __attribute__((always_inline)) {
    constexpr bool BORDER = decltype(border_tag)::value;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 hsc = *reinterpret_cast<const f32x4*>(lds + BN_OFF + 256 + 32 * j + 16 * lh);
      const f32x4 hsh = *reinterpret_cast<const f32x4*>(lds + BN_OFF + 384 + 32 * j + 16 * lh);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float u = acc[a][4 * j + e] * hsc[e] + hsh[e];
          v[e] = fmaxf(u, u * slope);
          if (BORDER) v[e] = (okbits & (1u << (16 + a))) ? v[e] : 0.f;  // conv2's zero padding: h is zero outside the image
        }
        uint2 hi, lo;
        split_bf16x4(v, hi, lo);
        if (VFX_B2D_ABL & 32) {
          asm volatile("" : : "v"(hi), "v"(lo));
          continue;
        }
        *reinterpret_cast<uint2*>(lds + hw[a] + 16 * j) = hi;
        *reinterpret_cast<uint2*>(lds + hw[a] + 16 * j + 64) = lo;
      }
    }
  };

#ifdef VFX_TIMING  // six stamps per tile and wave (the sixteen of VFX_TS_DECL cost registers this kernel does not have)
  unsigned long long ts_[6] = {};
#define B2D_TS(i) ts_[i] = __builtin_readcyclecounter()
#else
#define B2D_TS(i)
#endif
#ifdef VFX_TIMING  // block start / end on the chip-wide 100 MHz clock, behind the per-tile stamps: which blocks ran at the same time
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  begin_tile(t);
  __syncthreads();  // the bn2 table is visible (and nothing else has touched LDS yet)
  while (true) {
    B2D_TS(0);
    const bool border_now = border;
    // ---- x patch -> operand form in LDS ---------------------------------------------------------------------------------------------
    if (border_now) transform(std::true_type{});
    else transform(std::false_type{});
    B2D_TS(1);
    B2D_BARRIER();  // the patch is complete
    conv(a1, PITCH, 0, std::false_type{});
    B2D_TS(2);
    B2D_BARRIER();  // every wave is done reading the patch: h goes over it
    if (border_now) hwrite(std::true_type{});
    else hwrite(std::false_type{});
    B2D_TS(3);
    B2D_BARRIER();  // h is complete
    conv(a2, W1 * PIX, 9, std::true_type{});
    B2D_TS(4);
    // ---- y = conv2 + x: the tile is staged in LDS (pixel-major, PIX bytes per pixel) and leaves in full 128-byte lines ----------------------
    // (stored straight from the MFMA layout every instruction touches 32 lines with 32 bytes each: measured, profiles/r06_c40_block2d32_ablations.txt: the
    // stores alone were 0.12 ms of a 0.33 ms launch)
    B2D_BARRIER();  // every wave is done reading h
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(lds + sw[a] + 32 * j) = f32x4{acc[a][4 * j], acc[a][4 * j + 1], acc[a][4 * j + 2], acc[a][4 * j + 3]};
    B2D_BARRIER();  // the tile is staged
#pragma unroll
    for (int q = 0; q < NE; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(lds + srd + 32 * q * PIX) + res[q];
      if (VFX_B2D_ABL & 4) asm volatile("" : : "v"(v));
      else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry, (int)eoff(q, border_now), q * erow, 0);
      // A 16-byte store keeps reading its data registers after it has issued; the VALU instruction that forms the next piece in the
      // same registers must not be the very next one.  hipcc 7.2 knows this hazard for stores with an immediate soffset only and
      // leaves stores with a REGISTER soffset (these: q * erow) unprotected -- on gfx950 the first dword of the piece then went out
      // overwritten in the last lanes of each 16-lane group, 0.6 % of the pixels (scripts/b2d_check.py; profiles/r06_store_data_hazard.md).
      asm volatile("s_nop 1" ::: "memory");
    }
    B2D_TS(5);
#ifdef VFX_TIMING  // [tile][wave][16]: stamps 0, 3, 5, 7, 9, 12 of the 4-wave kernels' layout (scripts/phase_timing.py)
    if (p.timing && lane == 0) {
      unsigned long long* tp = p.timing + ((size_t)t * 4 + wave) * 16;
      tp[0] = ts_[0]; tp[1] = ts_[0]; tp[2] = ts_[0]; tp[3] = ts_[1]; tp[4] = ts_[1]; tp[5] = ts_[2]; tp[6] = ts_[2]; tp[7] = ts_[3];
      tp[8] = ts_[3]; tp[9] = ts_[4]; tp[10] = ts_[4]; tp[11] = ts_[4]; tp[12] = ts_[5];
    }
#endif
    wfirst = false;
    t += nblk;
    if (t >= t_end) {
#ifdef VFX_TIMING
      if (p.timing && tid == 0) {
        unsigned long long* tb = p.timing + (size_t)ntiles * 64 + (size_t)blockIdx.x * 4;
        tb[0] = rt0;
        tb[1] = __builtin_amdgcn_s_memrealtime();
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        tb[2] = ((unsigned long long)xcc << 32) | hw;
      }
#endif
      break;
    }
    begin_tile(t);    // the next tile's pixels are requested before the barrier (into the registers of the accumulators just stored)
    B2D_BARRIER();  // every wave is done reading h: the next patch goes over it
  }
}

bool block2d32_ok(const ResBlockParams& hp) {
  return hp.geo2d && hp.C == 32 && hp.tile_m == 256 && !hp.in1 && !hp.two_src && !hp.hionly && !(hp.tuning & VFX_TUNE_OLD_BLOCK2D) &&
         hp.recip_ok && hp.W <= 4096 && hp.H <= 65536;
}
// the 14 x 18 h grid (12 x 16 outputs) exists in this kernel only: plan_block2d may choose it where block2d32_ok() will hold
bool block2d32_has_tile(int TH, int W1) { return (TH == 16 && W1 == 16) || (TH == 14 && W1 == 18); }

template <int TH, int W1>
static void launch_b2d_t(const ResBlockParams* dparams, int64_t ntiles, hipStream_t stream) {
  using G = b2d::Geo<TH, W1>;
  static uint64_t attr_devices = 0;  // one static per instantiation
  static int ncu = 0;
  if (first_use_on_current_device(attr_devices)) {
    VFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_block2d32<TH, W1>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    int dev = 0;
    VFX_HIP(hipGetDevice(&dev));
    VFX_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
  }
  // three blocks per CU, a multiple of 8 (one contiguous tile range per XCD); never more blocks than tiles per XCD
  int64_t grid = (int64_t)(ncu > 0 ? ncu : 256) * VFX_B2D_PER_CU;
  grid = std::min<int64_t>(grid, ((ntiles + 7) / 8) * 8);
  grid = std::max<int64_t>(8, grid & ~(int64_t)7);
#ifdef VFX_B2D_FAKE_LDS  // measurement builds: a smaller LDS allocation than the kernel addresses (wrong results; occupancy experiments)
  hipLaunchKernelGGL((k_block2d32<TH, W1>), dim3((unsigned)grid), dim3(256), VFX_B2D_FAKE_LDS, stream, dparams, (int)ntiles);
#else
  hipLaunchKernelGGL((k_block2d32<TH, W1>), dim3((unsigned)grid), dim3(256), G::LDS_BYTES, stream, dparams, (int)ntiles);
#endif
}

void launch_block2d32(const ResBlockParams& hp, const ResBlockParams* dparams, hipStream_t stream) {
  const int64_t ntiles = (int64_t)hp.B * hp.tiles_h * hp.tiles_w;
  VFX_CHECK(ntiles > 0 && ntiles < ((int64_t)1 << 31), "block2d32: bad tile count");
  VFX_CHECK(block2d32_has_tile(hp.TH, hp.W1), "block2d32: no kernel for a %d x %d h grid", hp.TH, hp.W1);
  if (hp.TH == 14) launch_b2d_t<14, 18>(dparams, ntiles, stream);
  else launch_b2d_t<16, 16>(dparams, ntiles, stream);
  VFX_HIP(hipGetLastError());
}

}  // namespace vfx
