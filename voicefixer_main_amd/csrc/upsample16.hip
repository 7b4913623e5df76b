// upsample16.hip -- the TFGAN vocoder's ConvTranspose1d upsamplers in the 16-bit mode (fp16 operands, activated fp16 input, one fp16
// output) as a kernel of their own.
//
//     out[b, s q + r, :] = act( sum_t  W_r,t  x[b, q + dw(r, t), :]  + bias )        r < s output phases, two taps each
//
// (oracle/vocoder.py: F.conv_transpose1d(k = 2 s, stride s, padding s / 2 + s % 2, output_padding s % 2); the phases and their taps:
// vocoder.cpp, phase_taps).  k_conv runs this as a phased launch: one block per (spatial tile, phase, 128-cout range), each of
// which fetches the patch, walks Cin / 64 stages with a barrier each and stages its tile through LDS for the stores.  With K = 2 Cin
// that is a launch of prologues and epilogues: the 128 -> 64 upsampler (x3) ran at 0.16 of the MFMA peak and 1.9 TB/s, the 256 -> 128
// one at 0.30 and 1.9 TB/s (round 5, profiles/r05_pmc.txt) -- under both roofs.  Here
//   * a block owns TM input positions and ALL s x Cout outputs of them: the patch (TM + 2 rows x Cin channels of fp16) goes to LDS
//     ONCE by LDS-DMA (pre-swizzled, chunk-major: every 64-channel chunk is an image of 128-byte rows, the layout k_conv's 16-bit
//     stages read), ONE barrier, and then no wave waits for another again;
//   * a wave owns 64 positions x a stride of the (phase, 32-cout) blocks; per block it walks the (chunk, tap) groups -- four
//     K = 16 steps each, the order of k_conv's stage table, so the sums are the same sums -- with the weight fragments of three
//     groups in flight (global -> VGPR, fragment-packed as for k_conv: pack_conv mode 3), and stores the block straight from the
//     accumulators (v_permlane32_swap pairs, cf. resblock_w64.hip) through a descriptor of the CLIP: positions past a clip's own
//     end are out of range on the loads (zeros) and on the stores (dropped).
// Same products, same summation order, same rounding as the phased k_conv launch: bit-identical output.
#include "conv_common.h"
#include "vfx_internal.h"

namespace vfx {

// NCH: 64-channel chunks of the input (Cin / 64); TM: input positions per block
template <int NCH, int TM>
__global__ __launch_bounds__(256) void k_up16(const TapConvParams* __restrict__ pp) {
  constexpr int WAVES_N = TM == 128 ? 2 : 4;   // waves along the (phase, cout block) axis; the other 4 / WAVES_N along the positions
  constexpr int WM = 2;                        // 32-position blocks per wave
  constexpr int NROWS = TM + 2;                // patch rows: the union of the phases' taps is one position to each side
  constexpr int NG8 = (NROWS + 7) / 8;         // groups of 8 rows (one LDS-DMA instruction of a wave each)
  constexpr int PRB = NG8 * 8 * CROW;          // bytes per chunk image
  constexpr int G = 2 * NCH;                   // (chunk, tap) groups per cout block
  constexpr int RING = 4, AHEAD = 3;           // weight groups in registers / in flight (G % RING == 0: the slot of a group is static)
  constexpr unsigned kOob = 0x80000000u;
  static_assert(G % RING == 0 && (TM == 128 || TM == 64), "geometry");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* const lds = reinterpret_cast<char*>(smem);

  const TapConvParams& p = *pp;
  typedef const ConvStage VFX_CONST* StageTab;
  const StageTab stages = (StageTab)(uintptr_t)p.stages;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wn = wave % WAVES_N, wm = wave / WAVES_N;
  const int Tin = p.Wi, cph = p.cout_phase, nphase = p.nphase;
  const int NBc = cph >> 5, NB = nphase * NBc;
  const int tiles = (Tin + TM - 1) / TM;
  const int img = blockIdx.x / tiles, q0 = (blockIdx.x - img * tiles) * TM;
  const int Tn = p.lens ? min(Tin, ((const VFX_CONST int*)p.lens)[img] * p.lens_mul_in) : Tin;  // the clip's own end
  if (q0 >= Tn) return;
  constexpr int Cin = NCH * 64;
  const int64_t tstrideB = (int64_t)stages[0].tap_stride * 4;  // bytes between the weights of consecutive (chunk, tap) groups

  // ---- weight stream: group g = 2 c + t of cout block nb = wn + WAVES_N i ------------------------------------------------------------
  f32x4 Wr[RING][4];
  const int nblk = (NB - wn + WAVES_N - 1) / WAVES_N;  // cout blocks of this wave
  auto wbase = [&](int i) __attribute__((always_inline)) -> const char* {
    const int nb = wn + WAVES_N * i;
    const int r = nb / NBc, cb = nb - r * NBc;
    return reinterpret_cast<const char*>((const float*)stages[r * p.nstages].wt) + cb * 4096;
  };
  const unsigned lane16 = (unsigned)lane * 16u;
  auto fetch = [&](const char* wb, int g, f32x4 (&slot)[4]) __attribute__((always_inline)) {
    const char* a = wb + (int64_t)g * tstrideB + lane16;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) slot[ks] = *(const VFX_GLOBAL f32x4*)(a + ks * 1024);
  };
  const char* wcur = wbase(0);
#pragma unroll
  for (int g = 0; g < AHEAD; ++g) fetch(wcur, g, Wr[g]);  // in flight while the patch is staged

  // ---- the patch: rows q0 + dw_min .. + NROWS - 1 of every chunk, global -> LDS by DMA, pre-swizzled -----------------------------------
  {
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.seg[0].src) + (int64_t)img * Tin * (Cin * 2)), 0, Tn * (Cin * 2), 0x00020000);
    const int lrow = lane >> 3, cg = lane & 7;
    for (int g8 = wave; g8 < NG8; g8 += 4) {
      const int row = 8 * g8 + lrow;
      const int pos = q0 + p.dw_min + row;  // (in front of the clip: negative, wraps out of range; past its end: out of range)
      const unsigned o = row < NROWS ? (unsigned)pos * (unsigned)(Cin * 2) + (unsigned)((cg ^ ((row >> 1) & 7)) << 4) : kOob;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        VFX_LDS void* l = (VFX_LDS void*)(lds + c * PRB + g8 * 8 * CROW);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, l, 16, (int)(o + (unsigned)(c * 128)), 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
  __syncthreads();  // the only barrier of the block

  const unsigned obytes = (unsigned)(Tn * nphase * cph * 2);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(p.out_act) + (int64_t)img * Tin * nphase * cph * 2, 0, (int)obytes, 0x00020000);
  const float slope = p.act_slope;
  unsigned sat16 = 0;
  const int mrow = wm * 64 + l31;  // this lane's position inside the tile (block a: + 32 a)

  for (int i = 0; i < nblk; ++i) {
    const int nb = wn + WAVES_N * i;
    const int r = nb / NBc, cb = nb - r * NBc;
    const ConvStage VFX_CONST& S0 = stages[r * p.nstages];
    const char* wnext = i + 1 < nblk ? wbase(i + 1) : wcur;
    // patch rows of this lane's positions under the phase's two taps, and their swizzle keys
    int rb[2][WM], kx[2][WM];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int a = 0; a < WM; ++a) {
        const int row = mrow + 32 * a + (S0.poff[t] & 0xffff);
        rb[t][a] = row * CROW;
        kx[t][a] = (((row >> 1) & 7) << 4) ^ (16 * lh);
      }
    f32x16 acc[WM];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      // keep three groups in flight: group g + AHEAD of this block, or the first groups of the next one
      if (g + AHEAD < G) fetch(wcur, g + AHEAD, Wr[(g + AHEAD) % RING]);
      else if (i + 1 < nblk) fetch(wnext, g + AHEAD - G, Wr[(g + AHEAD) % RING]);
      const int c = g >> 1, t = g & 1;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        f16x8 px[WM];
#pragma unroll
        for (int a = 0; a < WM; ++a) px[a] = *reinterpret_cast<const f16x8*>(lds + c * PRB + rb[t][a] + (kx[t][a] ^ (32 * ks)));
        const f16x8 wf = __builtin_bit_cast(f16x8, Wr[g % RING][ks]);
#pragma unroll
        for (int a = 0; a < WM; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, px[a], acc[a], 0, 0, 0);
      }
    }
    wcur = wnext;

    // ---- the block's 64 positions x 32 couts: bias, activation, fp16, straight to memory ------------------------------------------------
    const int n = r * cph + cb * 32;
    f32x4 bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.bias) bv[j] = *(const VFX_GLOBAL f32x4*)(p.bias + n + 8 * j + 4 * lh);
    }
#pragma unroll
    for (int a = 0; a < WM; ++a) {
      const int q = q0 + mrow + 32 * a;
      const unsigned rowoff = (unsigned)q * (unsigned)(nphase * cph * 2) + (unsigned)(n * 2 + 16 * lh);  // (past the clip: out of range)
#pragma unroll
      for (int jp = 0; jp < 4; jp += 2) {
        unsigned q2[2][2];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          f32x4 u;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = acc[a][4 * (jp + rr) + e] + bv[jp + rr][e];
            u[e] = fmaxf(v, v * slope);
          }
          q2[rr][0] = pack_f16x2_sat16(u[0], u[1], true, sat16);
          q2[rr][1] = pack_f16x2_sat16(u[2], u[3], true, sat16);
        }
        // lanes 0-31 keep their run jp and receive the partner's run jp; lanes 32-63 receive the partner's run jp + 1 and keep theirs
        const auto s0 = __builtin_amdgcn_permlane32_swap(q2[0][0], q2[1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(q2[0][1], q2[1][1], false, false);
        const u32x4 w = {s0[0], s1[0], s0[1], s1[1]};
#ifdef VFX_ABL_NOSTORE  // timing-only build (wrong results): what the direct stores cost
        asm volatile("" : : "v"(w));
#else
        __builtin_amdgcn_raw_buffer_store_b128(w, ro, (int)(rowoff + (unsigned)(16 * jp)), 0, 0);
#endif
      }
    }
  }
  report_f16_saturation(f16_sat16_bad(sat16), p.flags);
}

// The phased launch `hp` (vocoder.cpp: one ConvTranspose1d) can run on k_up16: 16-bit mode, an activated fp16 source read in
// 64-channel stages (one per chunk, two taps each), one fp16 output and nothing else.
bool upsample16_ok(const TapConvParams& hp) {
  if (!(hp.split && hp.hionly && hp.nphase > 1 && hp.nseg == 1 && hp.seg[0].src_act && !hp.out && hp.out_act && !hp.residual &&
        !hp.residual_act && !hp.act_elu && !hp.act_scale && !hp.act_shift && !hp.per_tap && !hp.reflect_w && hp.ksplit <= 1))
    return false;
  if (hp.Hi != 1 || hp.Hg != 1 || hp.sh != 1 || hp.sw != 1 || hp.oh0 != 0 || hp.ow0 != 0 || hp.Wg != hp.Wi || hp.Wo != hp.Wi || hp.out_cmul) return false;
  const int Cin = hp.seg[0].C;
  if (Cin % 64 != 0 || hp.cout_phase % 32 != 0 || hp.nstages != Cin / 64) return false;
  // Cin = 128 and 256 (the x3 upsamplers of the recalled table): measured on one box against the phased k_conv launch
  // (profiles/r06_c7_upsamplers_per_launch.txt) 0.83 -> 0.48 ms and 0.55 -> 0.46 ms.  The wider ones are MFMA-bound launches of
  // one block per CU here (their patch fills the LDS) and lose by 2x (Cin = 512: 0.53 -> 1.01 ms, Cin = 1024: 0.29 -> 0.70 ms):
  // they stay on k_conv.
  const int nch = Cin / 64;
  if (nch != 2 && nch != 4) return false;
  if (hp.seg[0].ntaps != 3 || hp.dw_min != hp.seg[0].dw[0] || hp.seg[0].dw[2] - hp.seg[0].dw[0] != 2) return false;  // the union window: q - 1 .. q + 1
  if (hp.lens && hp.lens_mul_in != hp.lens_mul_out) return false;
  // 32-bit byte offsets inside one clip
  return (int64_t)hp.Wi * hp.nphase * hp.cout_phase * 2 < ((int64_t)1 << 30) && (int64_t)hp.Wi * Cin * 2 < ((int64_t)1 << 30);
}

template <int NCH, int TM>
static void launch_up16_t(const TapConvParams& hp, const TapConvParams* dparams, hipStream_t stream) {
  constexpr int NG8 = (TM + 2 + 7) / 8;
  const size_t lds = (size_t)NCH * NG8 * 8 * CROW;
  static uint64_t attr_devices = 0;
  if (first_use_on_current_device(attr_devices))
    VFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_up16<NCH, TM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t grid = (int64_t)hp.B * ((hp.Wi + TM - 1) / TM);
  VFX_CHECK(grid > 0 && grid < ((int64_t)1 << 31), "upsample16: bad grid");
  hipLaunchKernelGGL((k_up16<NCH, TM>), dim3((unsigned)grid), dim3(256), lds, stream, dparams);
}

void launch_upsample16(const TapConvParams& hp, const TapConvParams* dparams, hipStream_t stream) {
  VFX_CHECK(upsample16_ok(hp), "upsample16: not a launch this kernel runs");
  if (hp.seg[0].C == 128) launch_up16_t<2, 128>(hp, dparams, stream);
  else launch_up16_t<4, 128>(hp, dparams, stream);
  VFX_HIP(hipGetLastError());
}

}  // namespace vfx
