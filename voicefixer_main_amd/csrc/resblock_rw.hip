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
#include "conv_common.h"
#include "vfx_internal.h"

namespace vfx {

template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void k_resblock_rw(const ResBlockParams* __restrict__ pp, int ntiles, int per_block) {
  constexpr int C = 64;
  constexpr int NTHR = NW * 64;
  constexpr int MT = NW * 32;               // h positions per tile
  constexpr int PR = MT + 64;               // patch rows (plan_resblock)
  constexpr int RQ = NTHR / 16;             // rows per load group: 16 lanes x 16 bytes = one 256-byte row of raw x
  constexpr int NCQ = MT / RQ;              // centre loads per thread (8)
  constexpr int NHQ = 64 / RQ;              // halo loads per thread (2 or 4)
  constexpr int WM = 2;                     // 32-row MFMA blocks per wave: wave = 64 positions x 32 channels
  constexpr int ROWB = 128;                 // bytes per LDS row: 64 channels of fp16
  constexpr int H_OFF = PR * ROWB;          // h behind the patch (no barrier between conv1 and the h write)
  constexpr int LDO = C + 4;                // staged output row (floats), overlays patch + h
  static_assert(NCQ == 8 && MT * LDO * 4 <= (PR + MT) * ROWB, "staging must fit over patch + h");

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
  const int c0 = p.fold ? PW : d;  // patch row of h pixel 0: patch row m + c0 holds the input sample AT h pixel m (the residual)

  // ---- weights: this wave's 32 output channels of both convolutions, all taps, for the lifetime of the block ----------
  f16x8 W[2][2][3][2];  // [conv][32-channel chunk][tap][K = 16 step]
  {
    const int64_t ts = (int64_t)C * kKC;
    const unsigned nb_off = (unsigned)(wn * 1024 + lane * 4) * 4u;
#pragma unroll
    for (int cv = 0; cv < 2; ++cv)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const char* w = reinterpret_cast<const char*>((cv ? p.w2 : p.w1) + (3 * c + k) * ts) + nb_off;
          W[cv][c][k][0] = __builtin_bit_cast(f16x8, *(const VFX_GLOBAL f32x4*)(w));
          W[cv][c][k][1] = __builtin_bit_cast(f16x8, *(const VFX_GLOBAL f32x4*)(w + 2048));
        }
  }
  // conv1's bias lives in LDS behind patch + h (a VMEM load per tile would queue behind the prefetch), conv2's in 4 registers
  float* const b1s = reinterpret_cast<float*>(lds + (PR + MT) * ROWB);
  if (tid < C) b1s[tid] = p.b1[tid];
  const f32x4 b2v = *(const VFX_GLOBAL f32x4*)(p.b2 + 4 * cg);

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

  const int t_begin = blockIdx.x * per_block, t_end = min(t_begin + per_block, ntiles);
  auto tile_geom = [&](int t, int& img, int& j0, int& base_h) __attribute__((always_inline)) {
    const int tj = t % p.tiles_w;
    const int ti = (t / p.tiles_w) % p.tiles_h;
    img = t / tiles_per_img;
    j0 = tj * p.TWo;
    base_h = p.fold ? ti * TH * d + j0 - 1 : j0 - 1;
  };

  f32x4 PC[NCQ], PH[NHQ];  // the raw patch of the NEXT tile, in flight / landed
  auto request = [&](int t) __attribute__((always_inline)) {
    int img, j0, base_h;
    tile_geom(t, img, j0, base_h);
    const int base_x = base_h - d;
    const float* xi = p.x + (int64_t)img * T * C + 4 * cg;
#pragma unroll
    for (int q = 0; q < NCQ; ++q) {
      const int rel = rel_of(crow(q)), pos = base_x + rel;
      PC[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (rel >= 0 && (unsigned)pos < (unsigned)T) PC[q] = *(const VFX_GLOBAL f32x4*)(xi + (int64_t)pos * C);
    }
#pragma unroll
    for (int q = 0; q < NHQ; ++q) {
      const int rel = rel_of(hrow(q)), pos = base_x + rel;
      PH[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (rel >= 0 && (unsigned)pos < (unsigned)T) PH[q] = *(const VFX_GLOBAL f32x4*)(xi + (int64_t)pos * C);
    }
  };
  // raw row -> LeakyReLU -> fp16 -> this thread's 8 bytes of the patch row: piece cg >> 1 (8 channels) at slot piece ^ key
  auto to_patch = [&](const f32x4& raw, int pr, bool& sat) __attribute__((always_inline)) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(raw[e], raw[e] * slope);
    *reinterpret_cast<uint2*>(lds + pr * ROWB + (((cg >> 1) ^ ((pr >> 1) & 7)) << 4) + 8 * (cg & 1)) =
        make_uint2(pack_f16x2(v[0], v[1], sat), pack_f16x2(v[2], v[3], sat));
  };

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

  if (t_begin < t_end) request(t_begin);
  __syncthreads();  // b1s
  for (int t = t_begin; t < t_end; ++t) {
    asm volatile("" : "+v"(lr_v), "+v"(l31_v));
    int img, j0, base_h;
    tile_geom(t, img, j0, base_h);
    int arow1[WM], hrel_m[WM];  // this lane's h pixel: its patch row / its sample offset from h pixel (0, 0) (far outside: none)
#pragma unroll
    for (int a = 0; a < WM; ++a) {
      const int ml = (wm * WM + a) * 32 + l31_v;
      const int li = (int)(((unsigned)ml * inv_w1) >> 20), lj = ml - li * W1;
      arow1[a] = li < TH ? li * PW + lj : 0;
      hrel_m[a] = li < TH ? li * rowstride + lj : -(1 << 29);
    }

    // ---- the landed patch: to LDS as operands; its centre stays as the residual --------------------------------------
    f32x4 K[NCQ];
    {
      bool sat = false;
#pragma unroll
      for (int q = 0; q < NCQ; ++q) {
        K[q] = PC[q];
        if (crow(q) < P) to_patch(PC[q], crow(q), sat);
      }
#pragma unroll
      for (int q = 0; q < NHQ; ++q)
        if (hrow(q) < P) to_patch(PH[q], hrow(q), sat);
      report_f16_saturation(sat, p.flags);
    }
    __syncthreads();  // the patch is complete
    if (t + 1 < t_end) request(t + 1);  // lands while this tile is computed and stored

    // ---- conv1 (chunk-major taps: the order of k_resblock) ---------------------------------------------------------------
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        int rows[WM];
#pragma unroll
        for (int a = 0; a < WM; ++a) rows[a] = arow1[a] + p.poff[k];
        mma(W[0][c][k], lds, rows, c);
      }
    // ---- h = LeakyReLU(conv1 + b1) as fp16 operands, zero outside the sequence ------------------------------------------
    {
      bool sat = false;
#pragma unroll
      for (int a = 0; a < WM; ++a) {
        const int m = (wm * WM + a) * 32 + l31_v;
        const bool hval = (unsigned)(base_h + hrel_m[a]) < (unsigned)T;
        char* rowp = lds + H_OFF + m * ROWB + 8 * lh;
        const int key = (m >> 1) & 7;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4 u;
          const f32x4 b1v = *reinterpret_cast<const f32x4*>(b1s + wn * 32 + 8 * j + 4 * lh);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float tt = acc[a][4 * j + e] + b1v[e];
            u[e] = hval ? fmaxf(tt, tt * slope) : 0.f;
            acc[a][4 * j + e] = 0.f;
          }
          *reinterpret_cast<uint2*>(rowp + (((wn * 4 + j) ^ key) << 4)) = make_uint2(pack_f16x2(u[0], u[1], sat), pack_f16x2(u[2], u[3], sat));
        }
      }
      report_f16_saturation(sat, p.flags);
    }
    __syncthreads();  // h is complete
    // ---- conv2 from the resident h ----------------------------------------------------------------------------------------
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
        mma(W[1][c][k], lds + H_OFF, rows, c);
      }
    __syncthreads();  // every wave is done with the patch and h: the staged tile overlays them
    // ---- epilogue: stage the accumulators, then y = conv2 + x + b2 in the layout of the centre loads ------------------
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = (wm * WM + a) * 32 + l31_v;
        *reinterpret_cast<f32x4*>(smem + row * LDO + wn * 32 + 8 * j + 4 * lh) =
            f32x4{acc[a][4 * j], acc[a][4 * j + 1], acc[a][4 * j + 2], acc[a][4 * j + 3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[a][4 * j + e] = 0.f;
      }
    __syncthreads();
    {
      float* yi = p.y + (int64_t)img * T * C + 4 * cg;
      const bool even = (tid & 1) == 0;
      const float aslope = p.act_slope;
      bool sat = false;
#pragma unroll
      for (int q = 0; q < NCQ; ++q) {
        const int m = lr_v + RQ * q;
        const int li = (int)(((unsigned)m * inv_w1) >> 20), lj = m - li * W1;
        const int pos = base_h + li * rowstride + lj;
        const bool ok = (li < TH) & (lj >= 1) & (lj <= W1 - 2) & ((unsigned)pos < (unsigned)T) & (!p.fold | (j0 + lj - 1 < d));
        const f32x4 val = (*reinterpret_cast<const f32x4*>(smem + m * LDO + 4 * cg) + K[q]) + b2v;
        if (ok) *(VFX_GLOBAL f32x4*)(yi + (int64_t)pos * C) = val;
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
      if (p.ya) report_f16_saturation(sat, p.flags);
    }
    __syncthreads();  // the staged tile has been read: the next patch may overwrite it
  }
}

// h positions per tile of the register-weights kernel: 256 (default), 128 (VFX_RB_RW_MT=128), 0 = off (VFX_RB_RW=0: k_resblock)
int resblock_rw_tile() {
  static const int mt = [] {
    if (getenv("VFX_RB_RW") && atoi(getenv("VFX_RB_RW")) == 0) return 0;
    return (getenv("VFX_RB_RW_MT") && atoi(getenv("VFX_RB_RW_MT")) == 128) ? 128 : 256;
  }();
  return mt;
}

template <int NW>
static void launch_rw(const ResBlockParams* dparams, int64_t ntiles, hipStream_t stream) {
  constexpr int MT = NW * 32;
  const size_t lds = (size_t)(MT + 64 + MT) * 128 + 64 * sizeof(float);  // patch + h (the staged tile overlays them) + conv1's bias
  int dev = 0, cus = 256;
  VFX_HIP(hipGetDevice(&dev));
  VFX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int slots = cus * (NW == 4 ? 2 : 1);  // 256 registers per wave: 8 waves per CU
  const int per_block = (int)((ntiles + slots - 1) / slots);
  const int grid = (int)((ntiles + per_block - 1) / per_block);
  static uint64_t attr_devices = 0;
  if (first_use_on_current_device(attr_devices)) {
    VFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_resblock_rw<NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  hipLaunchKernelGGL((k_resblock_rw<NW>), dim3(grid), dim3(NW * 64), lds, stream, dparams, (int)ntiles, per_block);
}

void launch_resblock_rw(const ResBlockParams& hp, const ResBlockParams* dparams, hipStream_t stream) {
  VFX_CHECK(hp.rw && hp.hionly && hp.C == 64 && !hp.geo2d && !hp.asrc, "resblock_rw: needs the 16-bit mode and C = 64");
  const int64_t ntiles = (int64_t)hp.B * hp.tiles_h * hp.tiles_w;
  VFX_CHECK(ntiles > 0 && ntiles < ((int64_t)1 << 30), "resblock_rw: bad tile count");
  if (hp.tile_m == 256) launch_rw<8>(dparams, ntiles, stream);
  else if (hp.tile_m == 128) launch_rw<4>(dparams, ntiles, stream);
  else VFX_CHECK(false, "resblock_rw: tile of %d positions", hp.tile_m);
  VFX_HIP(hipGetLastError());
}

}  // namespace vfx
