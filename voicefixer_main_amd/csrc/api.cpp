// api.cpp -- handle, weight staging, arena and the extern "C" entry points of libvfx.so.
#include <cmath>
#include <cstring>
#include <cstdlib>

#include "vfx_internal.h"

namespace vfx {

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

// ---------------------------------------------------------------------------------------------
// device blob / arena planner
// ---------------------------------------------------------------------------------------------
void* DeviceBlob::alloc(size_t bytes) {
  void* p = nullptr;
  VFX_HIP(hipMalloc(&p, bytes ? bytes : 16));
  allocs.push_back(p);
  return p;
}
float* DeviceBlob::upload(const float* p, size_t n) {
  float* d = static_cast<float*>(alloc(n * sizeof(float)));
  if (n) VFX_HIP(hipMemcpy(d, p, n * sizeof(float), hipMemcpyHostToDevice));
  return d;
}
int* DeviceBlob::upload_i(const std::vector<int>& v) {
  int* d = static_cast<int*>(alloc(v.size() * sizeof(int)));
  if (!v.empty()) VFX_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice));
  return d;
}
void DeviceBlob::release() {
  for (void* p : allocs) (void)hipFree(p);
  allocs.clear();
}

size_t ArenaPlanner::alloc(size_t bytes) {
  bytes = (bytes + 255) & ~size_t(255);
  if (bytes == 0) bytes = 256;
  for (size_t i = 0; i < blocks.size(); ++i) {
    Block& b = blocks[i];
    if (b.free && b.size >= bytes) {
      if (b.size > bytes) {
        Block rest{b.off + bytes, b.size - bytes, true};
        b.size = bytes;
        b.free = false;
        const size_t off = b.off;
        blocks.insert(blocks.begin() + i + 1, rest);
        return off;
      }
      b.free = false;
      return b.off;
    }
  }
  // extend: merge with a trailing free block if there is one
  if (!blocks.empty() && blocks.back().free) {
    Block& b = blocks.back();
    b.size = bytes;
    b.free = false;
    high = b.off + bytes;
    return b.off;
  }
  blocks.push_back(Block{high, bytes, false});
  const size_t off = high;
  high += bytes;
  return off;
}

void ArenaPlanner::free(size_t off) {
  for (size_t i = 0; i < blocks.size(); ++i) {
    if (blocks[i].off == off && !blocks[i].free) {
      blocks[i].free = true;
      if (i + 1 < blocks.size() && blocks[i + 1].free) {
        blocks[i].size += blocks[i + 1].size;
        blocks.erase(blocks.begin() + i + 1);
      }
      if (i > 0 && blocks[i - 1].free) {
        blocks[i - 1].size += blocks[i].size;
        blocks.erase(blocks.begin() + i);
      }
      return;
    }
  }
  set_error("ArenaPlanner::free: unknown offset %zu", off);
  throw Error();
}

// ---------------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------------
// PyTorch Conv weight (Cout, CinTotal, KH, KW) -> [C/32][ntaps][Cout][32] for input channels
// [c_lo, c_lo + C); taps are (kh, kw) pairs.
static inline uint16_t bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// Rows -> MFMA fragment order.  Input: consecutive (chunk, tap) blocks of [Cout][32] floats.
// Output per block: [Cout/32][1024 floats]; inside a 1024-float cout block
//   split-bf16: 4 fragments (s, hl) = (k 0..15 | 16..31) x (hi | lo), each [64 lanes][8 bf16]:
//               lane l holds W[cout = 32*nb + (l & 31)][k = 16*s + 8*(l >> 5) + 0..7], w = hi + lo up
//               to 2^-17 relative;
//   fp32:       4 fragments g (k8 groups), each [64 lanes][4 floats]:
//               lane l holds W[cout = 32*nb + (l & 31)][k = 8*g + 4*(l >> 5) + 0..3].
// A wave reads one fragment with ONE coalesced 16-byte-per-lane load (conv.hip).
static inline uint16_t f16_rne(float f) {
  const float c = std::min(std::max(f, -65504.f), 65504.f);
  const _Float16 h = (_Float16)c;
  uint16_t u;
  memcpy(&u, &h, sizeof(u));
  return u;
}

// mode: 0 = fp32 fragments, 1 = split-bf16 (hi, lo), 2 = fp16 in the hi fragments (lo fragments zero: never loaded),
//       3 = fp16, 64-channel chunks (conv_chunk(mode) input channels per block): the four fragments of a cout block are
//           the K = 16 steps k 0..15 | 16..31 | 32..47 | 48..63, lane l holds W[cout = 32*nb + (l & 31)][k = 16*f + 8*(l >> 5) + 0..7]
//           -- the weights of a convolution whose source is an activated fp16 tensor (k_conv, H64)
int conv_chunk(int mode) { return mode == 3 ? 64 : kKC; }

bool& f16_weight_issue() {
  static thread_local bool issue = false;
  return issue;
}

void rows_to_fragments(std::vector<float>& packed, int Cout, int mode) {
  const bool split = mode != 0;
  if (mode == 2 || mode == 3) {
    // fp16 operands: f16_rne clamps and flushes silently, and no device flag sees a WEIGHT.  A tensor whose largest weight
    // is outside the fp16 range, or so deep in fp16's subnormal range (< 2^-17: fewer than 8 significant bits for the
    // LARGEST weight, less for the others) that the products lose the mode's accuracy, marks the weight set as "needs
    // strict arithmetic" (f16_weight_issue(); vocoder.cpp): every call on it raises VFX_FLAG_F16_SATURATED, so the
    // model-level calls re-run on split-bf16 operands and a raw caller sees the flag.  Measured
    // (tests/test_gpu_models.py): a tensor at 3e-5 (9 bits) still holds 55 dB, one at 3e-7 gives 18 dB.
    float wmax = 0.f;
    bool finite = true;
    for (float v : packed) {
      finite = finite && std::isfinite(v);
      wmax = std::max(wmax, std::fabs(v));
    }
    VFX_CHECK(finite, "precision 2: a convolution weight is not finite");
    if (wmax > 65504.f || (wmax != 0.f && wmax < 6.103515625e-05f / 8.f)) f16_weight_issue() = true;
  }
  const int kc = conv_chunk(mode);
  const size_t blk = (size_t)Cout * kc;         // input floats per (chunk, tap) block
  const size_t oblk = (size_t)Cout * kKC;       // output floats per block: Cout / 32 cout blocks of 1024 floats
  std::vector<float> res(packed.size() / blk * oblk);
  size_t oo = 0;
  for (size_t o = 0; o + blk <= packed.size(); o += blk, oo += oblk) {
    const float* in = &packed[o];
    for (int nb = 0; nb < Cout / 32; ++nb) {
      float* out = &res[oo + (size_t)nb * 1024];
      for (int f = 0; f < 4; ++f)
        for (int l = 0; l < 64; ++l) {
          const float* row = in + (size_t)(nb * 32 + (l & 31)) * kc;
          float* dst = out + (f * 64 + l) * 4;
          if (mode == 3) {
            uint16_t q[8];
            for (int j = 0; j < 8; ++j) q[j] = f16_rne(row[16 * f + 8 * (l >> 5) + j]);
            memcpy(dst, q, sizeof(q));
          } else if (split) {
            const int s2 = f >> 1, lo = f & 1;
            uint16_t q[8];
            for (int j = 0; j < 8; ++j) {
              const float v = row[16 * s2 + 8 * (l >> 5) + j];
              if (mode == 2) {
                q[j] = lo ? (uint16_t)0 : f16_rne(v);
              } else {
                const uint16_t hi = bf16_rne(v);
                q[j] = lo ? bf16_rne(v - bf16_to_f32(hi)) : hi;
              }
            }
            memcpy(dst, q, sizeof(q));
          } else {
            for (int e = 0; e < 4; ++e) dst[e] = row[8 * f + 4 * (l >> 5) + e];
          }
        }
    }
  }
  packed.swap(res);
}

std::vector<float> pack_conv(const float* w, int Cout, int CinTotal, int KH, int KW, int c_lo, int C,
                             const std::vector<std::pair<int, int>>& taps, int mode) {
  const int nt = (int)taps.size(), kc = conv_chunk(mode);
  VFX_CHECK(C % kc == 0, "pack_conv: %d input channels do not split into %d-channel chunks", C, kc);
  std::vector<float> out((size_t)C * nt * Cout);
  for (int ch = 0; ch < C / kc; ++ch)
    for (int t = 0; t < nt; ++t)
      for (int n = 0; n < Cout; ++n)
        for (int cc = 0; cc < kc; ++cc) {
          const int c = c_lo + ch * kc + cc;
          out[(((size_t)ch * nt + t) * Cout + n) * kc + cc] =
              w[(((size_t)n * CinTotal + c) * KH + taps[t].first) * KW + taps[t].second];
        }
  rows_to_fragments(out, Cout, mode);
  return out;
}

// PyTorch ConvTranspose weight (Cin, Cout, KH, KW) -> [Cin/chunk][ntaps][Cout][chunk] -> fragment order.
std::vector<float> pack_conv_transposed(const float* w, int Cin, int Cout, int KH, int KW,
                                        const std::vector<std::pair<int, int>>& taps, int mode) {
  const int nt = (int)taps.size(), kc = conv_chunk(mode);
  VFX_CHECK(Cin % kc == 0, "pack_conv_transposed: %d input channels do not split into %d-channel chunks", Cin, kc);
  std::vector<float> out((size_t)Cin * nt * Cout);
  for (int ch = 0; ch < Cin / kc; ++ch)
    for (int t = 0; t < nt; ++t)
      for (int n = 0; n < Cout; ++n)
        for (int cc = 0; cc < kc; ++cc) {
          const int c = ch * kc + cc;
          out[(((size_t)ch * nt + t) * Cout + n) * kc + cc] =
              w[(((size_t)c * Cout + n) * KH + taps[t].first) * KW + taps[t].second];
        }
  rows_to_fragments(out, Cout, mode);
  return out;
}

// Tile and patch geometry of a launch.  The tile is TH x TW <= 128 pixels of one image; if the
// bounding box of all taps around it fits kPatchMaxRows pixels, every (segment, chunk) is ONE stage
// reading all its taps from one patch; otherwise (Conv1d with dilation > 48) every (chunk, tap)
// is its own stage with a tile-sized patch.
static void plan_conv(TapConvParams& p) {
  int dh_lo = 1 << 30, dh_hi = -(1 << 30), dw_lo = 1 << 30, dw_hi = -(1 << 30);
  for (int s = 0; s < p.nseg; ++s)
    for (int t = 0; t < p.seg[s].ntaps; ++t) {
      dh_lo = std::min(dh_lo, p.seg[s].dh[t]);
      dh_hi = std::max(dh_hi, p.seg[s].dh[t]);
      dw_lo = std::min(dw_lo, p.seg[s].dw[t]);
      dw_hi = std::max(dw_hi, p.seg[s].dw[t]);
    }
  bool bodies_ok = true;  // conv.hip instantiates stage bodies for these tap counts only
  for (int s = 0; s < p.nseg; ++s) {
    const int nt = p.seg[s].ntaps;
    bodies_ok = bodies_ok && (nt == 1 || nt == 2 || nt == 3 || nt == 4 || nt == 7 || nt == 9);
  }
  // Tile shape: TW = 2^k columns x TH = min(128 / TW, Hg) rows.  Among the shapes whose all-taps window
  // fits kPatchMaxRows pixels take the one that wastes the fewest tile pixels on the image borders
  // (ties: the smaller window); if none fits, fall back to one stage per (chunk, tap) with a tile-sized patch.
  int best_shift = -1;
  double best_util = -1.0;
  int64_t best_P = 0;
  const int sft0 = p.Hg == 1 ? 7 : 0;
  // Rows of a TW-wide tile: 128 / TW, the image's height, and -- round 5 -- what keeps the all-taps window inside the patch
  // buffer.  A TALL NARROW image (level 6 of a 60-s segment: 188 x 3 pixels; the bottleneck: 94 x 1) had no shape at all whose
  // window fits at 128 / TW rows and fell back to one stage per (chunk, tap): nine times the stages, no split-K, 0.13 - 0.36 ms
  // per launch where a 16 x 10 s batch takes 0.05 (profiles/r05_1x60_vs_16x10_per_launch.txt).  Shapes that fitted before keep
  // their rows.
  // Round 6: an image NARROWER than the tile (level 6 of the mel ResUNet: 3 columns on 4-wide tiles) stages only the columns it has --
  // window width min(TW, Wg) + taps instead of TW + taps: 34 x 5 = 170 patch pixels hold all 32 rows of a 10-s clip's level 6 in ONE tile
  // (34 x 6 = 204 did not fit: two tiles of 30 + 2 rows, i.e. 1 536 blocks = two rounds of the chip's 768 slots per launch, 44-55 us
  // where one round takes 24-26).  The tile's dead columns read rows of the neighbouring patch pixels (the `dead` slack keeps them
  // inside the buffer) into accumulator columns nobody stores.
  auto win_w = [&](int TW) { return (int64_t)std::min(TW, p.Wg) + (int64_t)(dw_hi - dw_lo); };
  auto rows_of = [&](int sft) {
    const int TW = 1 << sft;
    const int64_t PW = win_w(TW), dead = TW - std::min(TW, p.Wg);
    const int64_t fit = (kPatchMaxRows - dead) / PW - (int64_t)(dh_hi - dh_lo);
    return (int)std::max<int64_t>(0, std::min<int64_t>(std::min(128 / TW, p.Hg), fit));
  };
  for (int sft = sft0; sft <= 7; ++sft) {
    const int TW = 1 << sft, TH = rows_of(sft);
    if (TW > 2 * p.Wg && sft > sft0) break;
    if (TH < 1) continue;
    const int64_t PH = TH + (int64_t)(dh_hi - dh_lo), PW = win_w(TW);
    if (PH * PW + (TW - std::min(TW, p.Wg)) > kPatchMaxRows || PW >= 65536) continue;
    const double covered = (double)((p.Hg + TH - 1) / TH) * ((p.Wg + TW - 1) / TW) * 128.0;
    const double util = (double)p.Hg * p.Wg / covered;
    if (util > best_util * 1.02 || (util > best_util * 0.98 && PH * PW < best_P)) {
      best_util = util;
      best_shift = sft;
      best_P = PH * PW;
    }
  }
  const bool window = best_shift >= 0 && bodies_ok;
  int tw_shift = best_shift;
  if (!window) {  // per-tap stages: any shape works, take the least wasteful one
    best_util = -1.0;
    for (int sft = sft0; sft <= 7; ++sft) {
      const int TW = 1 << sft, TH = std::min(128 / TW, p.Hg);
      if (TW > 2 * p.Wg && sft > sft0) break;
      const double covered = (double)((p.Hg + TH - 1) / TH) * ((p.Wg + TW - 1) / TW) * 128.0;
      const double util = (double)p.Hg * p.Wg / covered;
      if (util > best_util) {
        best_util = util;
        tw_shift = sft;
      }
    }
  }
  const int TW = 1 << tw_shift, TH = window ? rows_of(tw_shift) : std::min(128 / TW, p.Hg);
  p.TH = TH;
  p.TW = TW;
  p.tw_shift = tw_shift;
  p.tiles_h = (p.Hg + TH - 1) / TH;
  p.tiles_w = (p.Wg + TW - 1) / TW;
  if (window) {
    const int64_t PH = TH + (int64_t)(dh_hi - dh_lo);
    int64_t PW = win_w(TW);
    // An ODD patch width (the parity classes of a transposed 3x3 convolution: taps 0 / -1, window TW + 1) breaks what the 2-D
    // swizzle key rests on -- "the bank half of LDS row pi * PW + pj is pj & 1" (conv.hip) -- and every second fragment read of
    // those launches is a 2-way bank conflict (scripts/lds_conflicts_conv.py: 1.5 LDS cycles per lane group; PMC: 34-37 % conflict
    // cycles in the upsampler launches, review item 1d).  One unused column makes it even where it fits.
#ifndef VFX_ABL_ODD_PATCH_WIDTH  // (measurement builds keep the odd width)
    if (PH > 1 && (PW & 1) && PH * (PW + 1) + (TW - std::min(TW, p.Wg)) <= kPatchMaxRows) PW += 1;
#endif
    p.per_tap = 0;
    p.PW = (int)PW;
    p.P = (int)(PH * PW);
    p.dh_min = dh_lo;
    p.dw_min = dw_lo;
  } else {
    p.per_tap = 1;
    p.PW = TW;
    p.P = TH * TW;
    p.dh_min = p.dw_min = 0;
  }
}

// Channels per stage: 32, except for an activated source of the 16-bit mode -- an fp16 tensor whose 128-byte patch rows
// hold 64 channels (k_conv, H64).
int stage_channels(const TapConvParams& p, const TapSeg& S) { return (p.hionly && S.src_act) ? 64 : kKC; }

int count_stages(const TapConvParams& p) {  // per phase, for a phased launch
  int n = 0;
  for (int s = 0; s < p.nseg; ++s) n += (p.seg[s].C / stage_channels(p, p.seg[s])) * (p.per_tap ? p.seg[s].ntaps : 1);
  return n;
}

void build_stages(const TapConvParams& p, const float* ones, const float* zeros, ConvStage* out) {
  int k = 0;
  const int64_t tstride = (int64_t)(p.nphase > 1 ? p.cout_phase : p.Cout) * kKC;  // couts of ONE weight tensor
  for (int s = 0; s < p.nseg; ++s) {
    const TapSeg& S = p.seg[s];
    // per-tap launches run tap-major: the patch origin (and with it the kernel's cached pixel offsets)
    // then changes ntaps times per block instead of once per stage
    const int nwin = p.per_tap ? S.ntaps : 1;
    const int kc = stage_channels(p, S);
    const bool f16src = kc == 64;  // activated fp16 tensor: 2 bytes per element, a 64-channel chunk = 128 bytes = 32 floats
    for (int w = 0; w < nwin; ++w)
      for (int ch = 0; ch < S.C / kc; ++ch) {
        ConvStage st{};
        st.src = S.src + ch * kKC;  // 128 bytes per chunk in either form
        st.scale = (S.scale ? S.scale : ones) + (S.scale ? ch * kKC : 0);
        st.shift = (S.shift ? S.shift : zeros) + (S.shift ? ch * kKC : 0);
        st.C = f16src ? S.C / 2 : S.C;  // pixel stride in floats
        st.nbytes = (unsigned)((int64_t)p.B * p.in_img_stride * S.C * (f16src ? 2 : 4) - (int64_t)ch * kKC * 4);
        st.flags = S.src_act ? 1 : 0;
        if (S.src_act) {
          st.scale = ones;
          st.shift = zeros;
        }
        st.slope = S.act == ACT_NONE ? 1.f : S.slope;
        st.tap_stride = (int)tstride;
        if (p.per_tap) {
          st.wt = S.wt + ((int64_t)ch * S.ntaps + w) * tstride;
          st.ntaps = 1;
          st.dh0 = S.dh[w];
          st.dw0 = S.dw[w];
          st.poff[0] = 0;  // tile-sized patch, no shift
        } else {
          st.wt = S.wt + (int64_t)ch * S.ntaps * tstride;
          st.ntaps = S.ntaps;
          st.dh0 = p.dh_min;
          st.dw0 = p.dw_min;
          for (int t = 0; t < S.ntaps; ++t) {  // row offset | column shift << 16 | row shift << 24 (conv.hip, compute())
            const int dpi = S.dh[t] - p.dh_min, dpj = S.dw[t] - p.dw_min;
            VFX_CHECK(dpi < 128 && dpj < 256 && dpi * p.PW + dpj < 65536, "conv: tap offset out of range");
            st.poff[t] = (dpi * p.PW + dpj) | (dpj << 16) | (dpi << 24);
          }
        }
        out[k++] = st;
      }
  }
}

void set_conv1d_geometry(TapConvParams& p, int B, int T, int K, int dil, bool reflect) {
  p.B = B;
  // A dilation too wide for one patch (> 32 samples for k3) is folded: the sequence becomes an image with
  // rows of `dil` samples and the taps become vertical neighbours (TapConvParams, folded geometry).
  const bool fold = !reflect && (K - 1) * dil + 128 > kPatchMaxRows && dil >= 16;
  if (fold) {
    p.Hi = p.Hg = p.Ho = (T + dil - 1) / dil;
    p.Wi = p.Wg = p.Wo = dil;
    p.in_img_stride = p.in_limit = p.out_img_stride = p.out_limit = T;
  } else {
    p.Hi = p.Hg = p.Ho = 1;
    p.Wi = p.Wg = p.Wo = T;
  }
  p.sh = p.sw = 1;
  p.reflect_w = reflect ? 1 : 0;
  TapSeg& S = p.seg[0];
  S.ntaps = K;
  for (int k = 0; k < K; ++k) {
    S.dh[k] = fold ? k - K / 2 : 0;
    S.dw[k] = fold ? 0 : (k - K / 2) * dil;
  }
}

void finish_params(TapConvParams& p) {
  p.total_steps = 0;
  for (int s = 0; s < p.nseg; ++s) {
    VFX_CHECK(p.seg[s].C % stage_channels(p, p.seg[s]) == 0 && p.seg[s].ntaps >= 1 && p.seg[s].ntaps <= kMaxTaps,
              "conv: bad segment %d (C=%d ntaps=%d)", s, p.seg[s].C, p.seg[s].ntaps);
    VFX_CHECK(p.seg[s].C <= kIdentityLen, "conv: segment too wide for the identity tables");
    p.total_steps += p.seg[s].ntaps * (p.seg[s].C / stage_channels(p, p.seg[s]));
  }
  VFX_CHECK(!(p.hionly && p.out_act) || p.Cout % 8 == 0, "conv: an fp16 activated output needs Cout %% 8 == 0");
  VFX_CHECK((int64_t)p.Hi * p.Wi < (int64_t)1 << 31 && (int64_t)p.Ho * p.Wo < (int64_t)1 << 31, "conv: image too large");
  if (p.in_img_stride == 0) p.in_img_stride = p.in_limit = p.Hi * p.Wi;
  if (p.out_img_stride == 0) {
    p.out_img_stride = p.out_limit = p.Ho * p.Wo;
    p.M = p.B * p.Hg * p.Wg;
  } else {
    p.M = p.B * p.out_limit;  // folded 1-D launch: grid == output
  }
  VFX_CHECK((int64_t)p.B * p.Hg * p.Wg < (int64_t)1 << 31, "conv: too many output pixels");
  VFX_CHECK((int64_t)p.B * p.out_img_stride < (int64_t)1 << 31, "conv: too many output pixels");
  VFX_CHECK((int64_t)p.B * p.in_img_stride < (int64_t)1 << 31, "conv: too many input pixels");
  for (int s = 0; s < p.nseg; ++s)  // the kernel addresses a source with 32-bit byte offsets
    VFX_CHECK((int64_t)p.B * p.in_img_stride * p.seg[s].C * 4 < ((int64_t)1 << 32) - 4096,
              "conv: source tensor of segment %d exceeds 4 GiB", s);
  VFX_CHECK(p.out || p.out_act, "conv: no output");
  VFX_CHECK(!p.residual_act || (p.hionly && !p.residual && p.Cout % 4 == 0 && p.residual_inv_slope >= 1.f),
            "conv: an activated residual needs the 16-bit mode, no raw residual beside it and an invertible LeakyReLU");
  plan_conv(p);
  p.nstages = count_stages(p);
}

// ---------------------------------------------------------------------------------------------
// plans
// ---------------------------------------------------------------------------------------------
// Debug hooks.  vfx_config.tuning & VFX_TUNE_DEBUG_POISON_ARENA (per handle): the bytes of the arena the plan owns are set to NaN
// patterns before EVERY call (and when the arena grows), so that a kernel reading a workspace buffer nobody wrote shows up whatever
// ran before.  VFX_DEBUG_NAN in the environment, read ONCE per process: after every GEMM-shaped launch the outputs are scanned for
// non-finite values (synchronises; the first hit is reported on stderr).  Plan::run and the entry points do no getenv.
struct DebugSwitches {
  int debug_nan = 0;
  DebugSwitches() {
    if (const char* e = getenv("VFX_DEBUG_NAN")) debug_nan = atoi(e);
  }
};
static const DebugSwitches& debug_switches() {
  static const DebugSwitches s;
  return s;
}
// Called by the entry points before anything of the call is staged in the arena.
static void debug_poison(const vfx_handle* h, const Plan& plan, void* stream) {
  if ((h->cfg.tuning & VFX_TUNE_DEBUG_POISON_ARENA) && plan.bound_base && plan.arena_bytes)
    VFX_HIP(hipMemsetAsync(plan.bound_base, 0xFF, plan.arena_bytes, static_cast<hipStream_t>(stream)));
}

void Plan::run(const RunCtx& ctx) {
  if (debug_switches().debug_nan >= 2 && bound_base && arena_bytes) {
    // whole-arena scan after every op (tiny shapes only)
    for (size_t i = 0; i < ops.size(); ++i) {
      ops[i](ctx);
      const int64_t bad = count_nonfinite(reinterpret_cast<const float*>(bound_base), (int64_t)(arena_bytes / 4), ctx.stream);
      fprintf(stderr, "[vfx debug] after op %zu of %zu: %lld non-finite floats in the arena\n", i, ops.size(), (long long)bad);
    }
    return;
  }
  for (auto& f : ops) f(ctx);
}

static void debug_scan(const Plan* pl, const char* what, size_t idx, const float* rel, int64_t n, int M, int Cout, int K,
                       hipStream_t s) {
  if (!rel) return;
  const float* p = reinterpret_cast<const float*>(pl->bound_base + reinterpret_cast<size_t>(rel) - 1);
  const int64_t bad = count_nonfinite(p, n, s);
  if (bad) fprintf(stderr, "[vfx debug] %s #%zu (M=%d Cout=%d K=%d): %lld of %lld non-finite\n", what, idx, M, Cout, K,
                   (long long)bad, (long long)n);
}

// Algorithmic HBM bytes of a launch (SURVEY.md section 8d accounting: every tensor the launch must read or write,
// once): sources, residual, outputs; weights are L2-resident and not counted.
static double conv_algo_bytes(const TapConvParams& q) {
  const double in_px = (double)q.B * q.in_img_stride, out_px = (double)q.B * q.out_img_stride;
  double b = 0;
  if (q.nphase > 1) {
    b += in_px * q.seg[0].C * ((q.hionly && q.seg[0].src_act) ? 2.0 : 4.0);  // the phases share one source
  } else {
    for (int s2 = 0; s2 < q.nseg; ++s2) b += in_px * q.seg[s2].C * ((q.hionly && q.seg[s2].src_act) ? 2.0 : 4.0);
  }
  if (q.residual) b += out_px * q.Cout * 4.0;
  if (q.residual_act) b += out_px * q.Cout * 2.0;
  if (q.out) b += out_px * (q.out_cmul ? q.out_cmul : q.Cout) * 4.0;
  if (q.out_act) b += out_px * q.Cout * (q.hionly ? 2.0 : 4.0);
  return b;
}
// a ResStack layer's algorithmic bytes are x in + y out = 8 bytes per element and LAYER (a pair
// launch runs two layers).  What the kernel's own design moves on top of that (the fp16 forms xa / ya of the two-form trunk
// of the wide stacks; half of it for a pair, whose intermediate tensor never leaves the CU) is `resblock_design_bytes`.
// On the fp16 trunk of the 16-bit mode (round 4, ResBlockParams::x16) the tensors themselves are 2 bytes per element: x in + y
// out = 4 bytes per element and layer.
static double resblock_algo_bytes(const ResBlockParams& q) {
  const double n = (double)q.B * (q.geo2d ? (double)q.H * q.W : (double)q.T) * q.C;
  if (q.in1) return n * 4.0 + n / q.C * 4.0;  // one input channel in, y out
  if (q.two_src) return n * 12.0;                  // two sources in, y out
  return n * (q.x16 ? 4.0 : 8.0) * (q.dil2 > 0 ? 2.0 : 1.0);
}
static double resblock_design_bytes(const ResBlockParams& q) {
  const double n = (double)q.B * (q.geo2d ? (double)q.H * q.W : (double)q.T) * q.C;
  if (q.in1) return n * 4.0 + n / q.C * 4.0;
  if (q.two_src) return n * 12.0;
  if (q.x16) return n * 2.0 * ((q.x || q.xa ? 1.0 : 0.0) + (q.y ? 1.0 : 0.0) + (q.ya ? 1.0 : 0.0));  // one fp16 tensor in, y and / or ya out
  return n * 8.0 + (q.asrc ? n * 2.0 : 0.0) + (q.ya ? n * 2.0 : 0.0);  // x in, y out (+ the fp16 forms: xa in, ya out)
}

void PlanBuilder::add_conv(TapConvParams p) {
  p.split = h->cfg.precision != 0;
  p.tuning = h->cfg.tuning;
  p.short_clip = short_clip;
  finish_params(p);
  p.ksplit = (p.lens || no_splitk) ? 1 : choose_ksplit(p);  // (a launch with per-clip lengths skips the tiles past a clip's end)
  size_t ws_off = ~size_t(0);
  if (p.ksplit > 1) {  // the partial tiles live in the arena for the duration of this op
    ws_off = alloc_f((int64_t)p.ksplit * p.B * p.out_img_stride * p.Cout);
    p.ws = const_cast<float*>(rel_ptr(ws_off));
  } else {
    p.ksplit = 0;
  }
  const size_t idx = plan->host_params.size();
  plan->host_params.push_back(p);
  plan->conv_flops += conv_flops(p);
  plan->n_conv += 1;
  Plan* pl = plan;
  plan->ops.push_back([pl, idx](const RunCtx& c) {
    auto launch = [pl, idx](const TapConvParams& hp, const TapConvParams* dp, hipStream_t st) {
      launch_conv(hp, dp, st);
      if (hp.ksplit > 1) launch_splitk_reduce(pl->abs_params[idx], st);  // the same stream: ordered behind the partial tiles
    };
    if (c.prof && c.prof->enabled) {
      hipEvent_t a, b;
      VFX_HIP(hipEventCreate(&a));
      VFX_HIP(hipEventCreate(&b));
      VFX_HIP(hipEventRecord(a, c.stream));
      launch(pl->host_params[idx], pl->dev_params + idx, c.stream);
      VFX_HIP(hipEventRecord(b, c.stream));
      c.prof->events.push_back({a, b});
      c.prof->flops.push_back(conv_flops(pl->host_params[idx]));
      c.prof->bytes.push_back(conv_algo_bytes(pl->host_params[idx]));
      c.prof->design_bytes.push_back(conv_algo_bytes(pl->host_params[idx]));
      c.prof->bn.push_back(pl->host_params[idx].Cout);
      c.prof->desc.push_back(pl->host_params[idx]);
    } else {
      launch(pl->host_params[idx], pl->dev_params + idx, c.stream);
    }
    if (debug_switches().debug_nan) {
      const TapConvParams& q = pl->host_params[idx];
      int K = 0;
      for (int s2 = 0; s2 < q.nseg; ++s2) K += q.seg[s2].ntaps * q.seg[s2].C;
      const int64_t n = (int64_t)q.B * q.out_img_stride * (q.out_cmul ? q.out_cmul : q.Cout);
      debug_scan(pl, "conv out", idx, q.out, n, q.M, q.Cout, K, c.stream);
      debug_scan(pl, "conv out_act", idx, q.out_act, n, q.M, q.Cout, K, c.stream);
    }
  });
  if (ws_off != ~size_t(0)) free(ws_off);
}

void PlanBuilder::add_conv_phased(TapConvParams p, const std::vector<TapSeg>& phases) {
  VFX_CHECK(p.nseg == 1 && !phases.empty() && p.Cout % (int)phases.size() == 0, "phased conv: bad arguments");
  p.nphase = (int)phases.size();
  p.cout_phase = p.Cout / p.nphase;
  VFX_CHECK(p.cout_phase % 32 == 0, "phased conv: %d couts per phase", p.cout_phase);
  VFX_CHECK(!p.out_cmul || (p.out_cmul == p.cout_phase && p.nphase == (p.phase_rows ? 4 : 2) && p.sw == 2 && p.ow0 == 0 && !p.residual &&
                            !p.residual_act && !p.out_act && p.out && !p.bias),
            "phased conv: bad odd-width launch");
  VFX_CHECK(!p.phase_rows || (p.out_cmul && p.sh == 2 && p.oh0 == 0 && p.Wo >= 2), "phased conv: bad row-phased launch");
  const size_t idx = plan->host_params.size();
  plan->phase_segs[idx] = phases;
  add_conv(p);
  TapConvParams& hp = plan->host_params[idx];
  VFX_CHECK(!hp.per_tap, "phased conv: the union of the phases' taps does not fit one patch");
  // algorithmic work: every phase multiplies by its own taps only
  double k = 0;
  for (auto& S : phases) k += (double)S.ntaps * S.C;
  const double fl = 2.0 * (double)hp.M * hp.cout_phase * k;
  plan->conv_flops += fl - conv_flops(hp);
  hp.flops_override = fl;
  hp.up16 = (!(h->cfg.tuning & VFX_TUNE_NO_FUSED_UPSAMPLERS) && upsample16_ok(hp)) ? 1 : 0;
}

void PlanBuilder::add_resblock(ResBlockParams p) {
  p.tuning = h->cfg.tuning;
  if (p.geo2d) plan_block2d(p);
  else plan_resblock(p);
  const size_t idx = plan->host_rb.size();
  plan->host_rb.push_back(p);
  plan->conv_flops += resblock_flops(p);
  plan->n_conv += 1;
  Plan* pl = plan;
  plan->ops.push_back([pl, idx](const RunCtx& c) {
    const ResBlockParams& hp = pl->host_rb[idx];
    if (c.prof && c.prof->enabled) {
      hipEvent_t a, b;
      VFX_HIP(hipEventCreate(&a));
      VFX_HIP(hipEventCreate(&b));
      VFX_HIP(hipEventRecord(a, c.stream));
      launch_resblock(hp, pl->dev_rb + idx, c.stream);
      VFX_HIP(hipEventRecord(b, c.stream));
      c.prof->events.push_back({a, b});
      c.prof->flops.push_back(resblock_flops(hp));
      c.prof->bytes.push_back(resblock_algo_bytes(hp));
      c.prof->design_bytes.push_back(resblock_design_bytes(hp));
      c.prof->bn.push_back(hp.C);
      TapConvParams d{};
      d.M = hp.B * hp.T;
      d.Cout = hp.C;
      d.Wi = hp.dil;
      d.seg[0].C = hp.C;
      d.seg[0].ntaps = 6;
      d.hionly = hp.hionly;
      d.nstages = resblock_block_waves(hp);                                             // waves per block
      d.seg[0].ntaps = hp.dil2 > 0 ? 12 : (block2d32_ok(hp) ? 18 : 6);                  // pairs: four convolutions; 18: the persistent 2-D block
      c.prof->desc.push_back(d);
    } else {
      launch_resblock(hp, pl->dev_rb + idx, c.stream);
    }
    if (debug_switches().debug_nan && hp.y && !hp.x16)
      debug_scan(pl, "resblock y", idx, hp.y, (int64_t)hp.B * hp.T * hp.C, hp.B * hp.T, hp.C, 6 * hp.C, c.stream);
  });
}

static char* ensure_arena(vfx_handle* h, size_t bytes) {
  if (bytes <= h->arena_bytes) return h->arena;
  // A hipGraph captured from a plan (Plan::pinned) replays kernels whose parameter blocks hold ABSOLUTE pointers into the arena
  // as it was at capture time.  Growing the arena frees that memory: the next replay would read and write freed memory with no
  // error.  So a handle with a captured plan refuses to grow; the caller reserves the largest shape first (vfx_reserve), then
  // captures.
  for (auto& kv : h->plans)
    VFX_CHECK(!(kv.second->pinned && h->arena), "the workspace arena would have to grow from %zu to %zu bytes, but a hipGraph was captured from plan '%s' "
              "and replays kernels that point into the current arena: vfx_reserve() the largest (model, B, T) BEFORE capturing, "
              "or destroy the graph(s) and call vfx_unpin_plans(); if this call was itself being captured, that capture has failed", h->arena_bytes, bytes, kv.first.c_str());
  VFX_HIP(hipDeviceSynchronize());
  h->retired.clear();  // (evicted plans: their parameter blocks go now, the device is idle)
  if (h->arena) VFX_HIP(hipFree(h->arena));
  h->arena = nullptr;
  h->arena_bytes = 0;
  const size_t want = bytes + (bytes >> 4);
  void* p = nullptr;
  VFX_HIP(hipMalloc(&p, want));
  // Debug aid (tests): a freshly grown arena is filled with NaN patterns, so that a kernel reading a
  // workspace buffer before anything wrote it shows up as NaN instead of silently using stale values.
  if (h->cfg.tuning & VFX_TUNE_DEBUG_POISON_ARENA) VFX_HIP(hipMemset(p, 0xFF, want));
  h->arena = static_cast<char*>(p);
  h->arena_bytes = want;
  return h->arena;
}

// Rebase the plan's arena-relative pointers on the (possibly re-allocated) arena and upload
// the parameter blocks.
void bind_plan(vfx_handle* h, Plan& plan) {
  char* base = ensure_arena(h, plan.arena_bytes);
  if (plan.bound_base == base && (plan.dev_params || plan.dev_rb || (plan.host_params.empty() && plan.host_rb.empty()))) return;
  std::vector<TapConvParams> abs = plan.host_params;
  auto rebase = [&](const float* rel) -> const float* {
    return reinterpret_cast<const float*>(base + reinterpret_cast<size_t>(rel) - 1);
  };
  size_t total_stages = 0;
  for (auto& p : abs) total_stages += (size_t)p.nstages * std::max(p.nphase, 1);
  if (!plan.dev_stages && total_stages)
    plan.dev_stages = static_cast<ConvStage*>(plan.blob.alloc(total_stages * sizeof(ConvStage)));
  std::vector<ConvStage> stages(total_stages);
  size_t so = 0;
  for (auto& p : abs) {
    for (int s = 0; s < p.nseg; ++s) p.seg[s].src = rebase(p.seg[s].src);
    if (p.residual) p.residual = rebase(p.residual);
    if (p.residual_act) p.residual_act = rebase(p.residual_act);
    if (p.out) p.out = const_cast<float*>(rebase(p.out));
    if (p.out_act) p.out_act = const_cast<float*>(rebase(p.out_act));
    if (p.ws) p.ws = const_cast<float*>(rebase(p.ws));
    p.flags = h->d_flags;
    const size_t pidx = &p - abs.data();
    if (p.nphase > 1) {  // one stage table per phase, built from that phase's segment on the common patch geometry
      const std::vector<TapSeg>& segs = plan.phase_segs.at(pidx);
      for (int r = 0; r < p.nphase; ++r) {
        TapConvParams q = p;
        q.seg[0] = segs[r];
        q.seg[0].src = rebase(q.seg[0].src);
        build_stages(q, h->d_ones, h->d_zeros, stages.data() + so + (size_t)r * p.nstages);
      }
    } else {
      build_stages(p, h->d_ones, h->d_zeros, stages.data() + so);
    }
    p.stages = plan.dev_stages + so;
    so += (size_t)p.nstages * std::max(p.nphase, 1);
  }
  if (total_stages)
    VFX_HIP(hipMemcpy(plan.dev_stages, stages.data(), total_stages * sizeof(ConvStage), hipMemcpyHostToDevice));
  if (!plan.dev_params && !abs.empty())
    plan.dev_params = static_cast<TapConvParams*>(plan.blob.alloc(abs.size() * sizeof(TapConvParams)));
  if (!abs.empty())
    VFX_HIP(hipMemcpy(plan.dev_params, abs.data(), abs.size() * sizeof(TapConvParams), hipMemcpyHostToDevice));
  plan.abs_params = abs;  // host copy with absolute pointers (split-K reduce launches)
  if (!plan.host_rb.empty()) {
    std::vector<ResBlockParams> rb = plan.host_rb;
    for (auto& q : rb) {
      if (q.x) q.x = rebase(q.x);
      if (q.x2) q.x2 = rebase(q.x2);
      if (q.y) q.y = const_cast<float*>(rebase(q.y));
      if (q.xa) q.xa = rebase(q.xa);
      if (q.ya) q.ya = const_cast<float*>(rebase(q.ya));
      q.flags = h->d_flags;
    }
    if (!plan.dev_rb) plan.dev_rb = static_cast<ResBlockParams*>(plan.blob.alloc(rb.size() * sizeof(ResBlockParams)));
    VFX_HIP(hipMemcpy(plan.dev_rb, rb.data(), rb.size() * sizeof(ResBlockParams), hipMemcpyHostToDevice));
  }
  plan.bound_base = base;
}

// ---------------------------------------------------------------------------------------------
// front-end tables
// ---------------------------------------------------------------------------------------------
static double hz_to_mel(double f) { return 2595.0 * std::log10(1.0 + f / 700.0); }

void set_mel_filterbank(vfx_handle* h, const float* fb) {
  const int NB = h->cfg.n_fft / 2 + 1, NM = h->cfg.n_mels;
  std::vector<float> val;
  std::vector<int> start(NM), off(NM + 1);
  for (int m = 0; m < NM; ++m) {
    int lo = -1, hi = -1;
    for (int f = 0; f < NB; ++f)
      if (fb[(size_t)f * NM + m] != 0.f) {
        if (lo < 0) lo = f;
        hi = f;
      }
    off[m] = (int)val.size();
    start[m] = lo < 0 ? 0 : lo;
    if (lo >= 0)
      for (int f = lo; f <= hi; ++f) val.push_back(fb[(size_t)f * NM + m]);
  }
  off[NM] = (int)val.size();
  h->fe.fb_val = h->blob.upload(val);
  h->fe.fb_nnz = (int)val.size();
  h->fe.fb_start = h->blob.upload_i(start);
  h->fe.fb_off = h->blob.upload_i(off);
}

void init_front_end(vfx_handle* h) {
  const int N = h->cfg.n_fft;
  VFX_CHECK(N == 2048, "only n_fft = 2048 is supported (got %d)", N);
  VFX_CHECK(h->cfg.n_mels == 128, "only n_mels = 128 is supported (got %d)", h->cfg.n_mels);
  std::vector<float> win(N), tw(2 * (N / 2)), rtw(2 * (N / 2 + 1));
  for (int n = 0; n < N; ++n) win[n] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * n / N));
  for (int m = 0; m < N / 2; ++m) {
    tw[2 * m] = (float)std::cos(2.0 * M_PI * m / (N / 2));
    tw[2 * m + 1] = (float)(-std::sin(2.0 * M_PI * m / (N / 2)));
  }
  for (int k = 0; k <= N / 2; ++k) {
    rtw[2 * k] = (float)std::cos(2.0 * M_PI * k / N);
    rtw[2 * k + 1] = (float)(-std::sin(2.0 * M_PI * k / N));
  }
  h->fe.window = h->blob.upload(win);
  h->fe.twiddle = h->blob.upload(tw);
  h->fe.rtwiddle = h->blob.upload(rtw);

  // Default HTK mel filterbank (mel_scale.py:131-221) evaluated in double precision.  The
  // reference evaluates it with float32 torch ops; the Python shim therefore overrides this
  // table with the bit-identical one via vfx_load_tensor(VFX_MODEL_FRONTEND, "mel.fb").
  const int NB = N / 2 + 1, NM = h->cfg.n_mels;
  const double fmax = (double)(h->cfg.sample_rate / 2);
  std::vector<double> fpts(NM + 2);
  for (int i = 0; i < NM + 2; ++i) {
    const double m = hz_to_mel(0.0) + (hz_to_mel(fmax) - hz_to_mel(0.0)) * i / (NM + 1);
    fpts[i] = 700.0 * (std::pow(10.0, m / 2595.0) - 1.0);
  }
  std::vector<float> fb((size_t)NB * NM);
  for (int f = 0; f < NB; ++f) {
    const double hz = fmax * f / (NB - 1);
    for (int m = 0; m < NM; ++m) {
      const double up = (hz - fpts[m]) / (fpts[m + 1] - fpts[m]);
      const double down = (fpts[m + 2] - hz) / (fpts[m + 2] - fpts[m + 1]);
      fb[(size_t)f * NM + m] = (float)std::max(0.0, std::min(up, down));
    }
  }
  set_mel_filterbank(h, fb.data());

  // vocoder band weights: get_mel_weig (pytorch_util.py:141-155), base 10
  std::vector<float> invw(NM);
  const double norm0 = (fpts[2] - fpts[0]) / 2.0;
  for (int m = 0; m < NM; ++m) invw[m] = (float)(1.0 / (((fpts[m + 2] - fpts[m]) / 2.0) / norm0));
  h->fe.voc_inv_weight = h->blob.upload(invw);
}

bool stream_turns_enabled() {
  static const bool on = [] {
    const char* e = getenv("VFX_NO_STREAM_TURNS");
    return !(e && atoi(e) != 0);
  }();
  return on;
}

DeviceTurn& device_turn(int device) {
  static DeviceTurn turns[64];
  return turns[(unsigned)device % 64u];
}

}  // namespace vfx

// =============================================================================================
// extern "C"
// =============================================================================================
using namespace vfx;

#define VFX_API_BEGIN try {
// entry points that take a handle: NULL check + run on the handle's device, restore the caller's on exit
#define VFX_API_BEGIN_H(h) try { VFX_CHECK((h) != nullptr, "NULL handle"); ::vfx::DeviceGuard device_guard_((h)->device);
// ... and launch on a caller's stream: they take turns with calls on other streams of the device (StreamTurn, vfx_internal.h)
#define VFX_API_BEGIN_HS(h, stream) VFX_API_BEGIN_H(h) ::vfx::StreamTurn stream_turn_((h)->device, (stream));
#define VFX_API_END                         \
  }                                         \
  catch (const vfx::Error&) { return 1; }   \
  catch (const std::exception& e) {         \
    vfx::set_error("exception: %s", e.what()); \
    return 2;                               \
  }                                         \
  return 0;

extern "C" {

const char* vfx_last_error(void) { return vfx::g_err.c_str(); }

int vfx_default_config(vfx_config* cfg) {
  if (!cfg) return 1;
  std::memset(cfg, 0, sizeof(*cfg));
  cfg->sample_rate = 44100;
  cfg->n_fft = 2048;
  cfg->hop = 441;
  cfg->n_mels = 128;
  cfg->voc_cond_channels = 512;
  cfg->voc_cond_layers = 5;
  cfg->voc_channels = 1024;
  cfg->voc_n_stages = 4;
  const int scales[4] = {7, 7, 3, 3};
  for (int i = 0; i < 4; ++i) {
    cfg->voc_scales[i] = scales[i];
    cfg->voc_depth[i] = 8;
  }
  cfg->voc_dilation_base = 3;
  cfg->voc_min_db = -115.f;
  cfg->voc_amp_floor = 1e-5f;
  cfg->voc_norm_range = 4.f;
  cfg->voc_up_slope = 0.2f;
  cfg->voc_res_slope = 0.01f;
  cfg->precision = 1;
  return 0;
}

int vfx_create(int device, const vfx_config* cfg, vfx_handle** out) {
  VFX_API_BEGIN
  VFX_CHECK(out != nullptr, "vfx_create: out is NULL");
  DeviceGuard device_guard_(device);  // the caller's current device is left as it was
  auto h = std::make_unique<vfx_handle>();
  h->device = device;
  if (cfg) h->cfg = *cfg; else vfx_default_config(&h->cfg);
  VFX_CHECK(h->cfg.voc_n_stages >= 1 && h->cfg.voc_n_stages <= VFX_MAX_STAGES, "bad voc_n_stages");
  VFX_CHECK((h->cfg.tuning & ~4095) == 0, "vfx_create: unknown bits in vfx_config.tuning (0x%x)", h->cfg.tuning);
  if (h->cfg.tuning) {  // never silent: a non-default kernel selection is announced
    static const char* names[] = {"NO_FUSED_STACKS", "NO_FUSED_WIDE", "NO_FUSED_UNET", "NO_PERSISTENT_C64", "NO_PAIRS", "NO_SPLITK",
                                  "F32_TRUNK", "SMALL_2D_TILES", "DEBUG_POISON_ARENA", "NO_FUSED_UPSAMPLERS", "OLD_BLOCK2D", "TWO_LAUNCH_UPSAMPLERS"};
    std::string msg;
    for (int b = 0; b < 12; ++b)
      if (h->cfg.tuning & (1 << b)) msg += std::string(msg.empty() ? "" : " | ") + "VFX_TUNE_" + names[b];
    fprintf(stderr, "[libvfx] handle on device %d uses non-default kernel selection: tuning = 0x%x (%s)\n", device, h->cfg.tuning,
            msg.c_str());
  }
  init_front_end(h.get());
  {
    std::vector<float> ones(kIdentityLen, 1.f), zeros(kIdentityLen, 0.f);
    h->d_ones = h->blob.upload(ones);
    h->d_zeros = h->blob.upload(zeros);
  }
  h->d_flags = static_cast<int*>(h->blob.alloc(sizeof(int)));
  VFX_HIP(hipMemset(h->d_flags, 0, sizeof(int)));
  h->d_lens = static_cast<int*>(h->blob.alloc(9 * kMaxVarlenClips * sizeof(int)));  // rows 0-2: the batch; 3-5: the ResUNet group, 6-8: the vocoder run in flight
  VFX_HIP(hipMemset(h->d_lens, 0, 9 * kMaxVarlenClips * sizeof(int)));
  *out = h.release();
  VFX_API_END
}

int vfx_destroy(vfx_handle* h) {
  if (!h) return 0;
  int prev = -1;
  if (hipGetDevice(&prev) != hipSuccess) prev = -1;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  h->plans.clear();
  h->retired.clear();
  if (h->arena) (void)hipFree(h->arena);
  if (h->scratch) (void)hipFree(h->scratch);
  delete h;
  if (prev >= 0) (void)hipSetDevice(prev);
  return 0;
}

int vfx_load_tensor(vfx_handle* h, int model, const char* name, const float* data, const int64_t* shape, int ndim) {
  VFX_API_BEGIN
  VFX_CHECK(h && name && data, "vfx_load_tensor: NULL argument");
  VFX_CHECK(model >= 0 && model <= VFX_MODEL_FRONTEND, "vfx_load_tensor: bad model id %d", model);
  HostTensor t;
  int64_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    t.shape.push_back(shape[i]);
    n *= shape[i];
  }
  t.data.assign(data, data + n);
  h->staged[model][name] = std::move(t);
  VFX_API_END
}

int vfx_finalize_weights(vfx_handle* h, int model) {
  VFX_API_BEGIN_H(h)
  h->plans.clear();
  if (model == VFX_MODEL_UNET_MEL || model == VFX_MODEL_UNET_SPEC) {
    h->unet[model] = build_unet_weights(h, model);
  } else if (model == VFX_MODEL_VOCODER) {
    h->voc = build_vocoder_weights(h);
  } else if (model == VFX_MODEL_FRONTEND) {
    auto it = h->staged[model].find("mel.fb");
    if (it != h->staged[model].end()) {
      VFX_CHECK(it->second.shape.size() == 2 && it->second.shape[0] == h->cfg.n_fft / 2 + 1 &&
                    it->second.shape[1] == h->cfg.n_mels,
                "mel.fb must be (%d, %d)", h->cfg.n_fft / 2 + 1, h->cfg.n_mels);
      VFX_HIP(hipDeviceSynchronize());
      set_mel_filterbank(h, it->second.data.data());
    }
  } else {
    VFX_CHECK(false, "vfx_finalize_weights: bad model id %d", model);
  }
  h->staged[model].clear();
  VFX_API_END
}

int vfx_take_flags(vfx_handle* h, void* stream, int* flags_out) {
  VFX_API_BEGIN_H(h)
  VFX_CHECK(h && flags_out, "NULL argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int v = 0;
  VFX_HIP(hipMemcpyAsync(&v, h->d_flags, sizeof(int), hipMemcpyDeviceToHost, s));
  VFX_HIP(hipMemsetAsync(h->d_flags, 0, sizeof(int), s));
  VFX_HIP(hipStreamSynchronize(s));
  *flags_out = v;
  VFX_API_END
}

int vfx_unpin_plans(vfx_handle* h) {
  VFX_API_BEGIN_H(h)
  VFX_CHECK(h, "NULL handle");
  for (auto& kv : h->plans) kv.second->pinned = false;
  VFX_API_END
}

// Read-and-clear only the bits in `mask`: the other sticky bits stay raised for whoever checks them later (a deferred
// saturation check of the vocoder must survive the UNet stage's negative-input check, models.VoiceFixer.forward).
int vfx_take_flags_masked(vfx_handle* h, void* stream, int mask, int* flags_out) {
  VFX_API_BEGIN_H(h)
  VFX_CHECK(h && flags_out, "NULL argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int v = 0;
  VFX_HIP(hipMemcpyAsync(&v, h->d_flags, sizeof(int), hipMemcpyDeviceToHost, s));
  VFX_HIP(hipMemsetAsync(h->d_flags, 0, sizeof(int), s));
  VFX_HIP(hipStreamSynchronize(s));
  if (v & ~mask) launch_or_flags(h->d_flags, v & ~mask, s);  // in stream order, before anything the caller enqueues next
  *flags_out = v & mask;
  VFX_API_END
}

// The two halves of a turn for work this library does not enqueue itself: the replay of a hipGraph captured from its calls
// (a captured call is exempt from the turns, so its replay would otherwise overlap a live call of another stream).
int vfx_turn_begin(int device, void* stream) {
  VFX_API_BEGIN
  DeviceGuard device_guard_(device);
  if (stream_turns_enabled()) {
    DeviceTurn& d = device_turn(device);
    std::lock_guard<std::mutex> lock(d.mu);
    turn_wait(d, static_cast<hipStream_t>(stream));
  }
  VFX_API_END
}

int vfx_turn_end(int device, void* stream) {
  VFX_API_BEGIN
  DeviceGuard device_guard_(device);
  if (stream_turns_enabled()) {
    DeviceTurn& d = device_turn(device);
    std::lock_guard<std::mutex> lock(d.mu);
    VFX_CHECK(turn_record(d, static_cast<hipStream_t>(stream)), "vfx_turn_end: cannot record the end of the turn on this stream");
  }
  VFX_API_END
}

// ---------------------------------------------------------------------------------------------
// front-end
// ---------------------------------------------------------------------------------------------
static int frames_of(const vfx_handle* h, int L) { return L / h->cfg.hop + 1; }

int vfx_stft_mel(vfx_handle* h, const float* wav, int B, int L, float* mel, float* sp, float* cosp, float* sinp,
                 int log10_mel, void* stream) {
  VFX_API_BEGIN_HS(h, stream)
  VFX_CHECK(h && wav, "NULL argument");
  VFX_CHECK(B > 0 && L > h->cfg.n_fft / 2, "vfx_stft_mel: need B > 0 and L > n_fft/2 (reflect padding), got B=%d L=%d", B, L);
  launch_stft_mel(h->fe, wav, B, L, frames_of(h, L), mel, sp, cosp, sinp, log10_mel, h->cfg.hop, 1e-8f,
                  static_cast<hipStream_t>(stream));
  VFX_API_END
}

int vfx_stft_phase(vfx_handle* h, const float* wav, int B, int L, float* sp, float* cosp, float* sinp, float eps,
                   void* stream) {
  VFX_API_BEGIN_HS(h, stream)
  VFX_CHECK(wav && (sp || cosp || sinp), "NULL argument");
  VFX_CHECK(B > 0 && L > h->cfg.n_fft / 2, "vfx_stft_phase: need B > 0 and L > n_fft/2 (reflect padding), got B=%d L=%d", B, L);
  VFX_CHECK(eps >= 0.f, "vfx_stft_phase: eps must be >= 0 (got %g)", (double)eps);
  launch_stft_mel(h->fe, wav, B, L, frames_of(h, L), nullptr, sp, cosp, sinp, 0, h->cfg.hop, eps,
                  static_cast<hipStream_t>(stream));
  VFX_API_END
}

int vfx_mel_project(vfx_handle* h, const float* sp, int64_t rows, float* mel, void* stream) {
  VFX_API_BEGIN_HS(h, stream)
  VFX_CHECK(h && sp && mel && rows > 0, "bad argument");
  launch_mel_project(h->fe, sp, rows, mel, static_cast<hipStream_t>(stream));
  VFX_API_END
}

int vfx_spectral_metrics(vfx_handle* h, const float* est, const float* target, int B, int T, int F, float* out, void* stream) {
  VFX_API_BEGIN_HS(h, stream)
  VFX_CHECK(h && est && target && out && B > 0 && T > 0 && F > 0 && B <= 65535, "bad argument");
  // per-frame partial sums live in the arena (no plan is running concurrently: single stream, single thread)
  Plan tmp;
  tmp.arena_bytes = (size_t)B * T * 4 * sizeof(double);
  if (tmp.arena_bytes > h->arena_bytes) h->plans.clear();  // plans hold absolute pointers into the old arena
  bind_plan(h, tmp);
  launch_spectral_metrics(est, target, B, T, F, reinterpret_cast<double*>(h->arena), out, static_cast<hipStream_t>(stream));
  VFX_API_END
}

int vfx_chunk_gather(vfx_handle* h, const float* x, int B, int L, int win, int hop, int lead, int n_chunks,
                     float* chunks, void* stream) {
  VFX_API_BEGIN_HS(h, stream)
  VFX_CHECK(h && x && chunks && B > 0 && L > 0 && win > 0 && hop > 0 && lead >= 0 && n_chunks > 0, "bad argument");
  VFX_CHECK(B <= 65535 && n_chunks <= 65535, "chunk grid too large");
  launch_chunk_gather(x, B, L, win, hop, lead, n_chunks, chunks, static_cast<hipStream_t>(stream));
  VFX_API_END
}

int vfx_chunk_ola(vfx_handle* h, const float* frames, const float* window, float scale, int B, int n_chunks, int win,
                  int hop, int lead, int L, float* y, void* stream) {
  VFX_API_BEGIN_HS(h, stream)
  VFX_CHECK(h && frames && y && B > 0 && L > 0 && win > 0 && hop > 0 && lead >= 0 && n_chunks > 0, "bad argument");
  VFX_CHECK(B <= 65535, "batch too large");
  launch_chunk_ola(frames, window, scale, B, n_chunks, win, hop, lead, L, y, static_cast<hipStream_t>(stream));
  VFX_API_END
}

int vfx_istft(vfx_handle* h, const float* re, const float* im, int B, int T, int L, float* wav, void* stream) {
  VFX_API_BEGIN_HS(h, stream)
  VFX_CHECK(h && re && im && wav && B > 0 && T > 0 && L > 0, "bad argument");
  launch_istft(h->fe, re, im, B, T, L, h->cfg.hop, wav, static_cast<hipStream_t>(stream));
  VFX_API_END
}

// ---------------------------------------------------------------------------------------------
// model stages
// ---------------------------------------------------------------------------------------------
static std::shared_ptr<Plan> get_plan(vfx_handle* h, const std::string& key,
                                      const std::function<void(PlanBuilder&)>& build, void* stream = nullptr) {
  auto it = h->plans.find(key);
  std::shared_ptr<Plan> plan;
  if (it == h->plans.end()) {
    plan = std::make_shared<Plan>();
    PlanBuilder pb{h, plan.get(), {}};
    build(pb);
    plan->arena_bytes = pb.arena.high;
    // bounded cache: drop the least recently used plan(s) first (their parameter blocks are hipFree'd, which waits
    // for the device: nothing in flight still reads them).  Plans a hipGraph was captured from are never dropped: the
    // graph's kernel nodes keep the plan's device parameter blocks as arguments (the cache then grows past the bound).
    while (h->plans.size() >= kMaxCachedPlans) {
      auto victim = h->plans.end();
      for (auto i = h->plans.begin(); i != h->plans.end(); ++i)
        if (!i->second->pinned && (victim == h->plans.end() || i->second->last_use < victim->second->last_use)) victim = i;
      if (victim == h->plans.end()) break;
      // the victim's parameter blocks are hipFree'd when the Plan dies, and hipFree waits for the whole device: retire it instead and
      // let the plans go in batches (every 64 evictions, when the arena grows -- both wait for the device anyway -- and at
      // vfx_destroy), so that a test set with more distinct shapes than the cache holds does not stall the GPU once per call
      h->retired.push_back(victim->second);
      h->plans.erase(victim);
      if (h->retired.size() >= 64) h->retired.clear();
    }
    h->plans[key] = plan;
  } else {
    plan = it->second;
  }
  bool capturing = false;
  if (stream) {  // the legacy (NULL) stream cannot be captured
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    capturing = hipStreamIsCapturing(static_cast<hipStream_t>(stream), &st) == hipSuccess && st == hipStreamCaptureStatusActive;
  }
  plan->last_use = ++h->plan_tick;
  // bind first, pin afterwards: when the arena would have to grow under a capture (or under an older pinned plan)
  // ensure_arena throws, the capture fails in the caller -- and a plan no graph was captured from must not stay pinned, it
  // would refuse every later growth until somebody finds vfx_unpin_plans()
  bind_plan(h, *plan);
  if (capturing) plan->pinned = true;
  return plan;
}

static std::string key_of(const char* tag, int B, int T, int x = 0) {
  char buf[96];
  snprintf(buf, sizeof(buf), "%s:%d:%d:%d", tag, B, T, x);
  return buf;
}

static BufRef ext(int slot) {
  BufRef b;
  b.ext = true;
  b.slot = slot;
  return b;
}
static BufRef arena_buf(size_t off) {
  BufRef b;
  b.off = off;
  return b;
}

size_t vfx_workspace_bytes(vfx_handle* h, int model, int B, int T) {
  try {
    if (!h) return 0;
    DeviceGuard device_guard_(h->device);
    Plan plan;
    PlanBuilder pb{h, &plan, {}};
    if (model == VFX_MODEL_UNET_MEL) build_unet_mel(pb, B, T, ext(0), ext(1));
    else if (model == VFX_MODEL_UNET_SPEC) build_unet_spec(pb, B, T, ext(0), ext(1), ext(2), ext(3), ext(4));
    else if (model == VFX_MODEL_VOCODER) build_vocoder(pb, B, T, ext(0), ext(1));
    else return 0;
    return pb.arena.high;
  } catch (...) {
    return 0;
  }
}

int vfx_reserve(vfx_handle* h, int model, int B, int T) {
  VFX_API_BEGIN_H(h)
  const size_t need = vfx_workspace_bytes(h, model, B, T);
  VFX_CHECK(need > 0, "vfx_reserve: cannot plan model %d (weights finalized?): %s", model, vfx_last_error());
  Plan tmp;
  tmp.arena_bytes = need;
  bind_plan(h, tmp);
  VFX_API_END
}

// The convolution kernels address a tensor with 32-bit byte offsets (buffer descriptors, LDS-DMA), so one
// launch handles tensors below 4 GiB.  A batch whose largest activation would exceed that is run as
// consecutive sub-batches on the same stream (clips are independent; each sub-batch re-uses the cached plan).
// Largest activation per clip: ResUNet level 1 (Tpad x W x 32 ch) and the vocoder's widest-in-bytes stack.
static int max_clips_per_launch(const vfx_handle* h, int T, bool unet_mel, bool unet_spec, bool voc) {
  const int64_t Tpad = (T + 63) / 64 * 64;
  int64_t per_clip = 1;
  if (unet_mel) per_clip = std::max<int64_t>(per_clip, Tpad * 127 * 64 * 4);      // cat(up, skip) of 2 x 32 channels
  if (unet_spec) per_clip = std::max<int64_t>(per_clip, Tpad * 1024 * 64 * 4);
  if (voc) {
    int64_t len = T + T % 2 + 4, c = h->cfg.voc_channels;
    per_clip = std::max(per_clip, len * c * 4);
    for (int i = 0; i < h->cfg.voc_n_stages; ++i) {
      len *= h->cfg.voc_scales[i];
      c /= 2;
      per_clip = std::max(per_clip, len * c * 4);
    }
  }
  int64_t lim = (((int64_t)1 << 32) - (1 << 20)) / per_clip;
  if (const char* e = getenv("VFX_MAX_CLIPS")) lim = std::min<int64_t>(lim, atoi(e));  // tests: force the sub-batch path
  return (int)std::max<int64_t>(1, std::min<int64_t>(lim, 1 << 20));
}

static int vfx_resunet_mel_1(vfx_handle* h, const float* mel_linear, int B, int T, float* logmel_out, void* stream);
int vfx_resunet_mel(vfx_handle* h, const float* mel_linear, int B, int T, float* logmel_out, void* stream) {
  if (!h || B <= 0 || T <= 0) return vfx_resunet_mel_1(h, mel_linear, B, T, logmel_out, stream);
  const int step = max_clips_per_launch(h, T, true, false, false);
  for (int b = 0; b < B; b += step) {
    const int rc = vfx_resunet_mel_1(h, mel_linear + (int64_t)b * T * 128, std::min(step, B - b), T,
                                     logmel_out + (int64_t)b * T * 128, stream);
    if (rc) return rc;
  }
  return 0;
}
static int vfx_resunet_mel_1(vfx_handle* h, const float* mel_linear, int B, int T, float* logmel_out, void* stream) {
  VFX_API_BEGIN_HS(h, stream)
  VFX_CHECK(h && mel_linear && logmel_out && B > 0 && T > 0, "bad argument");
  VFX_CHECK(h->unet[VFX_MODEL_UNET_MEL], "vfx_resunet_mel: weights of the mel ResUNet are not finalized");
  auto plan = get_plan(h, key_of("unet_mel", B, T),
                       [&](PlanBuilder& pb) { build_unet_mel(pb, B, T, ext(0), ext(1)); }, stream);
  debug_poison(h, *plan, stream);
  RunCtx ctx{static_cast<hipStream_t>(stream), {const_cast<float*>(mel_linear), logmel_out}, h->d_flags, &h->prof};
  plan->run(ctx);
  VFX_API_END
}

static int vfx_resunet_spec_1(vfx_handle* h, const float* sp, const float* wav, int B, int T, int L, float* wav_out,
                              void* stream);
int vfx_resunet_spec(vfx_handle* h, const float* sp, const float* wav, int B, int T, int L, float* wav_out,
                     void* stream) {
  if (!h || B <= 0 || T <= 0) return vfx_resunet_spec_1(h, sp, wav, B, T, L, wav_out, stream);
  const int step = max_clips_per_launch(h, T, false, true, false);
  const int64_t nb = h->cfg.n_fft / 2 + 1;
  for (int b = 0; b < B; b += step) {
    const int rc = vfx_resunet_spec_1(h, sp + (int64_t)b * T * nb, wav + (int64_t)b * L, std::min(step, B - b), T, L,
                                      wav_out + (int64_t)b * L, stream);
    if (rc) return rc;
  }
  return 0;
}
static int vfx_resunet_spec_1(vfx_handle* h, const float* sp, const float* wav, int B, int T, int L, float* wav_out,
                              void* stream) {
  VFX_API_BEGIN_HS(h, stream)
  VFX_CHECK(h && sp && wav && wav_out && B > 0 && T > 0, "bad argument");
  VFX_CHECK(h->unet[VFX_MODEL_UNET_SPEC], "vfx_resunet_spec: weights of the spectrogram ResUNet are not finalized");
  VFX_CHECK(T == frames_of(h, L), "vfx_resunet_spec: T=%d does not match L=%d (expected %d frames)", T, L, frames_of(h, L));
  const size_t nsp = (size_t)B * T * (h->cfg.n_fft / 2 + 1);
  auto plan = get_plan(h, key_of("unet_spec", B, T), [&](PlanBuilder& pb) {
    auto& nm = pb.plan->named;
    nm["cos"] = pb.alloc_f(nsp);
    nm["sin"] = pb.alloc_f(nsp);
    nm["re"] = pb.alloc_f(nsp);
    nm["im"] = pb.alloc_f(nsp);
    build_unet_spec(pb, B, T, ext(0), arena_buf(nm["cos"]), arena_buf(nm["sin"]), arena_buf(nm["re"]), arena_buf(nm["im"]));
  }, stream);
  debug_poison(h, *plan, stream);
  const size_t off_cos = plan->named["cos"], off_sin = plan->named["sin"], off_re = plan->named["re"],
               off_im = plan->named["im"];
  hipStream_t s = static_cast<hipStream_t>(stream);
  char* base = h->arena;
  float* cosb = reinterpret_cast<float*>(base + off_cos);
  float* sinb = reinterpret_cast<float*>(base + off_sin);
  // second STFT of the same audio for the phase (unet_v2.py:96)
  launch_stft_mel(h->fe, wav, B, L, T, nullptr, nullptr, cosb, sinb, 0, h->cfg.hop, 1e-8f, s);
  RunCtx ctx{s, {const_cast<float*>(sp)}, h->d_flags, &h->prof};
  plan->run(ctx);
  launch_istft(h->fe, reinterpret_cast<float*>(base + off_re), reinterpret_cast<float*>(base + off_im), B, T, L,
               h->cfg.hop, wav_out, s);
  VFX_API_END
}

int64_t vfx_vocoder_out_len(vfx_handle* h, int T) { return h ? vocoder_out_len(h->cfg, T) : -1; }

static int vfx_vocoder_1(vfx_handle* h, const float* mel_linear, int B, int T, float* wav_out, void* stream);
int vfx_vocoder(vfx_handle* h, const float* mel_linear, int B, int T, float* wav_out, void* stream) {
  if (!h || B <= 0 || T <= 0) return vfx_vocoder_1(h, mel_linear, B, T, wav_out, stream);
  const int step = max_clips_per_launch(h, T, false, false, true);
  const int64_t Llong = vocoder_out_len(h->cfg, T);
  for (int b = 0; b < B; b += step) {
    const int rc = vfx_vocoder_1(h, mel_linear + (int64_t)b * T * 128, std::min(step, B - b), T, wav_out + (int64_t)b * Llong, stream);
    if (rc) return rc;
  }
  return 0;
}
static int vfx_vocoder_1(vfx_handle* h, const float* mel_linear, int B, int T, float* wav_out, void* stream) {
  VFX_API_BEGIN_HS(h, stream)
  VFX_CHECK(h && mel_linear && wav_out && B > 0 && T > 0, "bad argument");
  VFX_CHECK(h->voc, "vfx_vocoder: vocoder weights are not finalized");
  auto plan = get_plan(h, key_of("vocoder", B, T), [&](PlanBuilder& pb) { build_vocoder(pb, B, T, ext(0), ext(1)); }, stream);
  debug_poison(h, *plan, stream);
  RunCtx ctx{static_cast<hipStream_t>(stream), {const_cast<float*>(mel_linear), wav_out}, h->d_flags, &h->prof};
  plan->run(ctx);
  VFX_API_END
}

static int vfx_restore_gsr_1(vfx_handle* h, const float* wav, int B, int L, float* wav_out, float* logmel_out, int flags,
                             void* stream);
int vfx_restore_gsr(vfx_handle* h, const float* wav, int B, int L, float* wav_out, float* logmel_out, int flags,
                    void* stream) {
  if (!h || B <= 0 || L <= 0) return vfx_restore_gsr_1(h, wav, B, L, wav_out, logmel_out, flags, stream);
  const int T = L / h->cfg.hop + 1;
  const int step = max_clips_per_launch(h, T, true, false, true);
  for (int b = 0; b < B; b += step) {
    const int rc = vfx_restore_gsr_1(h, wav + (int64_t)b * L, std::min(step, B - b), L, wav_out + (int64_t)b * L,
                                     logmel_out ? logmel_out + (int64_t)b * T * 128 : nullptr, flags, stream);
    if (rc) return rc;
  }
  return 0;
}
static int vfx_restore_gsr_1(vfx_handle* h, const float* wav, int B, int L, float* wav_out, float* logmel_out, int flags,
                             void* stream) {
  VFX_API_BEGIN_HS(h, stream)
  VFX_CHECK(h && wav && wav_out && B > 0, "bad argument");
  VFX_CHECK(L > h->cfg.n_fft / 2, "vfx_restore_gsr: clip too short for reflect padding (L=%d)", L);
  VFX_CHECK(h->unet[VFX_MODEL_UNET_MEL] && h->voc, "vfx_restore_gsr: weights are not finalized");
  const int T = frames_of(h, L);
  const int64_t Llong = vocoder_out_len(h->cfg, T);
  const int unify = flags & 1;
  // keyed on L, not on T: the plan's launches hold the sample count (STFT row stride and reflection point, trim_center) -- two
  // clips with the same frame count and different lengths must not share it (rounds 1-4 keyed on T: the second of two such
  // clips on one handle was framed and trimmed with the first one's length; found by the varlen comparison of round 5)
  auto plan = get_plan(h, key_of("restore_gsr", B, L, unify), [&](PlanBuilder& pb) {
    const int64_t nmel = (int64_t)B * T * 128;
    const size_t o_mel = pb.alloc_f(nmel), o_log = pb.alloc_f(nmel), o_den = pb.alloc_f(nmel);
    const size_t o_long = pb.alloc_f((int64_t)B * Llong), o_ws = pb.alloc_f(2 * B + 64), o_pk = pb.alloc_f(B + 64);
    vfx_handle* hh = pb.h;
    Plan* pl = pb.plan;
    // pre(): STFT -> magnitude -> mel (eval_gsr_voicefixer.py:19-25)
    pl->ops.push_back([=](const RunCtx& c) {
      launch_stft_mel(hh->fe, c.ext[0], B, L, T, reinterpret_cast<float*>(pl->bound_base + o_mel), nullptr, nullptr,
                      nullptr, 0, hh->cfg.hop, 1e-8f, c.stream);
    });
    build_unet_mel(pb, B, T, arena_buf(o_mel), arena_buf(o_log));
    pl->ops.push_back([=](const RunCtx& c) {
      float* lg = reinterpret_cast<float*>(pl->bound_base + o_log);
      if (c.ext[2]) VFX_HIP(hipMemcpyAsync(c.ext[2], lg, sizeof(float) * nmel, hipMemcpyDeviceToDevice, c.stream));
      launch_from_log(lg, reinterpret_cast<float*>(pl->bound_base + o_mel), B, T, unify,
                      reinterpret_cast<float*>(pl->bound_base + o_ws), reinterpret_cast<float*>(pl->bound_base + o_den),
                      c.stream);
    });
    const BufRef peak_buf = arena_buf(o_pk);  // per-clip peak, produced by the vocoder tail
    build_vocoder(pb, B, T, arena_buf(o_den), arena_buf(o_long), &peak_buf);
    pl->ops.push_back([=](const RunCtx& c) {
      launch_peak_trim(reinterpret_cast<float*>(pl->bound_base + o_long), B, Llong, L,
                       reinterpret_cast<float*>(pl->bound_base + o_pk), /*have_peak=*/true, c.ext[1], c.stream, c.flags);
    });
  }, stream);
  debug_poison(h, *plan, stream);
  RunCtx ctx{static_cast<hipStream_t>(stream), {const_cast<float*>(wav), wav_out, logmel_out}, h->d_flags, &h->prof};
  plan->run(ctx);
  VFX_API_END
}

// A batch of clips of UNEQUAL length through the same per-segment body (the reference restores one file per call, of any
// length: evaluation_proc/eval.py:119-134, eval_gsr_voicefixer.py:47-74).  wav (B, Lmax): clip b = the first lengths[b] samples
// of row b.  Every clip gets what its own batch-of-one call computes:
//   * STFT: frames and the reflection at ITS end (fDomainHelper.py:26-28, center = True framing);
//   * ResUNet: all clips of a call share the padded frame count 64 * ceil(T_b / 64) -- a REQUIREMENT of this entry point (the
//     caller buckets by it) -- and a clip's rows past T_b are zeros like the network's own time padding (unet.py:75-77);
//   * vocoder: every launch stops the clip at its own length (zero padding, the k7 reflections, the tail of -4 frames);
//   * peak normalisation and trim_center per clip; wav_out (B, Lmax) and logmel_out (B, Tmax, 128) are zero past a clip's end.
static int vfx_restore_gsr_varlen_1(vfx_handle* h, const float* wav, int B, int Lmax, const int* lengths, float* wav_out,
                                    float* logmel_out, int flags, void* stream);
int vfx_restore_gsr_varlen(vfx_handle* h, const float* wav, int B, int Lmax, const int* lengths, float* wav_out,
                           float* logmel_out, int flags, void* stream) {
  if (!h || B <= 0 || Lmax <= 0 || !lengths)
    return vfx_restore_gsr_varlen_1(h, wav, B, Lmax, lengths, wav_out, logmel_out, flags, stream);
  const int T = Lmax / h->cfg.hop + 1;
  // (the launches of one call are sub-batched inside it: the ResUNet per padded frame count, the vocoder per run of clips)
  const int step = kMaxVarlenClips;
  for (int b = 0; b < B; b += step) {
    const int rc = vfx_restore_gsr_varlen_1(h, wav + (int64_t)b * Lmax, std::min(step, B - b), Lmax, lengths + b,
                                            wav_out + (int64_t)b * Lmax, logmel_out ? logmel_out + (int64_t)b * T * 128 : nullptr,
                                            flags, stream);
    if (rc) return rc;
  }
  return 0;
}
// Handle-owned scratch beside the arena (grow-only): the tensors that travel BETWEEN the plans of one varlen call (every plan
// places its own buffers from offset 0 of the arena).  Growing frees and re-allocates: the device is idle then (hipFree waits).
static char* ensure_scratch(vfx_handle* h, size_t bytes) {
  if (bytes <= h->scratch_bytes) return h->scratch;
  for (auto& kv : h->plans)
    VFX_CHECK(!kv.second->pinned, "the varlen scratch would have to grow from %zu to %zu bytes, but a hipGraph was captured from plan '%s': "
              "run the largest varlen batch once BEFORE capturing", h->scratch_bytes, bytes, kv.first.c_str());
  VFX_HIP(hipDeviceSynchronize());
  if (h->scratch) VFX_HIP(hipFree(h->scratch));
  h->scratch = nullptr;
  h->scratch_bytes = 0;
  const size_t want = bytes + (bytes >> 3);
  void* p = nullptr;
  VFX_HIP(hipMalloc(&p, want));
  h->scratch = static_cast<char*>(p);
  h->scratch_bytes = want;
  return h->scratch;
}

// Clips of one ResUNet launch of a varlen call: a group's clip count is rounded up (dummy clips of zero frames) so that a test set
// meets a handful of (count, padded frames) shapes instead of one per group size -- a plan is 15 ms of host work to build
static int padded_group(int n) { return n <= 2 ? n : (n <= 8 ? (n + 1) / 2 * 2 : (n + 3) / 4 * 4); }

static int vfx_restore_gsr_varlen_1(vfx_handle* h, const float* wav, int B, int Lmax, const int* lengths, float* wav_out,
                                    float* logmel_out, int flags, void* stream) {
  VFX_API_BEGIN_HS(h, stream)
  VFX_CHECK(h && wav && wav_out && lengths && B > 0 && B <= kMaxVarlenClips, "bad argument");
  VFX_CHECK(h->unet[VFX_MODEL_UNET_MEL] && h->voc, "vfx_restore_gsr_varlen: weights are not finalized");
  const int hop = h->cfg.hop;
  const int T = frames_of(h, Lmax);
  std::vector<int> host(3 * (size_t)B);
  std::map<int, std::vector<int>, std::greater<int>> groups;  // padded frame count -> clips (longest group first)
  for (int b = 0; b < B; ++b) {
    const int Lb = lengths[b];
    VFX_CHECK(Lb > h->cfg.n_fft / 2 && Lb <= Lmax, "vfx_restore_gsr_varlen: clip %d has %d samples (need %d < length <= Lmax = %d)", b, Lb,
              h->cfg.n_fft / 2, Lmax);
    const int Tb = Lb / hop + 1;
    host[b] = Lb;
    host[B + b] = Tb;
    host[2 * (size_t)B + b] = Tb + Tb % 2 + 4;
    groups[(Tb + 63) / 64 * 64].push_back(b);
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  // the per-clip lengths of THIS call, in stream order behind the kernels of the previous one: rows 0-2 = samples, frames, vocoder
  // frames of the batch; rows 3-4 = frames and batch index of the clips of the ResUNet group in flight; rows 6-8 = samples, frames,
  // vocoder frames of the vocoder run in flight
  constexpr int cap = kMaxVarlenClips;
  int* const d_l = h->d_lens;
  int* const d_t = h->d_lens + cap;
  int* const d_gt = h->d_lens + 3 * cap;
  int* const d_gi = h->d_lens + 4 * cap;
  int* const d_vl = h->d_lens + 6 * cap;
  int* const d_vt = h->d_lens + 7 * cap;
  int* const d_vtp = h->d_lens + 8 * cap;
  launch_set_lens(h->d_lens, cap, host.data(), B, s);
  const int unify = flags & 1;
  // ---- vocoder runs: consecutive clips (the callers hand them over sorted by length), each run on a compact (n, Tv, 128) tensor whose
  // frame count Tv is the run's own longest clip, rounded up to a multiple of 64 frames (a handful of plan shapes per test set)
  struct Run { int b0, n, Tv; };
  std::vector<Run> runs;
  for (int b0 = 0; b0 < B;) {
    int n = 0, tmax = 0;
    while (b0 + n < B) {
      const int tb = host[B + b0 + n];
      const int tnew = std::min(T, (std::max(tmax, tb) + 63) / 64 * 64);
      if (n > 0 && n + 1 > max_clips_per_launch(h, tnew, false, false, true)) break;
      // (measured, 128 clips of 2-8 s: cutting a run when its padding passes 7 % -- ~20 clips per run instead of ~43 -- changes
      // nothing: 0.777 -> 0.770 of an equal-length batch; what the mixed set loses is the ResUNet's deep levels on ~13 clips per group)
      tmax = tnew;
      ++n;
    }
    runs.push_back({b0, n, tmax});
    b0 += n;
  }
  // ---- the tensors between the stages (scratch): linear mel, log-mel estimate, restored linear mel of the batch; one ResUNet group's
  // compact input and output; one vocoder run's compact mel, long waveform, energy sums and peaks
  const int64_t nmel = (int64_t)B * T * 128;
  int64_t gmax = 0, vmel = 0, vlong = 0;
  int nmax = 0;
  for (auto& kv : groups) {
    const int step = std::max(1, max_clips_per_launch(h, kv.first, true, false, false) / 4 * 4);
    gmax = std::max<int64_t>(gmax, (int64_t)std::min(padded_group((int)kv.second.size()), step) * kv.first * 128);
  }
  for (auto& r : runs) {
    vmel = std::max<int64_t>(vmel, (int64_t)r.n * r.Tv * 128);
    vlong = std::max<int64_t>(vlong, (int64_t)r.n * vocoder_out_len(h->cfg, r.Tv));
    nmax = std::max(nmax, r.n);
  }
  auto up = [](int64_t n) { return (size_t)((n + 63) / 64 * 64) * sizeof(float); };
  const size_t o_mel = 0, o_log = o_mel + up(nmel), o_den = o_log + up(nmel), o_gin = o_den + up(nmel), o_gout = o_gin + up(gmax),
               o_vmel = o_gout + up(gmax), o_long = o_vmel + up(vmel), o_ws = o_long + up(vlong), o_pk = o_ws + up(2 * B + 64),
               o_end = o_pk + up(nmax + 64);
  char* const sc = ensure_scratch(h, o_end);
  float* const mel = reinterpret_cast<float*>(sc + o_mel);
  float* const lg = reinterpret_cast<float*>(sc + o_log);
  float* const den = reinterpret_cast<float*>(sc + o_den);
  float* const gin = reinterpret_cast<float*>(sc + o_gin);
  float* const gout = reinterpret_cast<float*>(sc + o_gout);
  float* const vm = reinterpret_cast<float*>(sc + o_vmel);
  float* const wlong = reinterpret_cast<float*>(sc + o_long);
  float* const ws = reinterpret_cast<float*>(sc + o_ws);
  float* const pk = reinterpret_cast<float*>(sc + o_pk);
  // ---- pre(): STFT -> magnitude -> mel, every clip framed and reflected at its own length (eval_gsr_voicefixer.py:19-25)
  launch_stft_mel(h->fe, wav, B, Lmax, T, mel, nullptr, nullptr, nullptr, 0, hop, 1e-8f, s, d_l);
  // ---- the mel ResUNet, one launch set per padded frame count (unet.py:75-77 pads every clip to ITS multiple of 64 frames): the
  // group's clips -- wherever they sit in the batch -- are gathered into a compact (Bg, Tpad, 128) tensor, restored, scattered back
  for (auto& kv : groups) {
    const int Tg = kv.first;
    const std::vector<int>& idx = kv.second;
    const int step = std::max(1, max_clips_per_launch(h, Tg, true, false, false) / 4 * 4);
    for (size_t at = 0; at < idx.size(); at += step) {
      const int n = (int)std::min<size_t>(step, idx.size() - at);
      const int np = std::min(padded_group(n), std::max(step, n));
      std::vector<int> hg(3 * (size_t)np, 0);
      for (int j = 0; j < np; ++j) {
        hg[j] = j < n ? host[B + idx[at + j]] : 0;                    // frames (a dummy clip: none -- all rows are padding)
        hg[(size_t)np + j] = idx[at + std::min(j, n - 1)];            // batch index
      }
      launch_set_lens(d_gt, cap, hg.data(), np, s);
      launch_gather_rows(mel, d_gi, gin, np, T, Tg, 128, s);
      auto plan = get_plan(h, key_of("unet_mel_vg", np, Tg), [&](PlanBuilder& pb) {
        pb.lens_t = d_gt;
        build_unet_mel(pb, np, Tg, ext(0), ext(1));
      }, stream);
      debug_poison(h, *plan, stream);
      RunCtx ctx{s, {gin, gout}, h->d_flags, &h->prof};
      plan->run(ctx);
      launch_scatter_rows(gout, d_gi, lg, n, T, Tg, 128, s);
    }
  }
  if (logmel_out) launch_copy_rows_masked(lg, logmel_out, B, T, 128, d_t, s);
  launch_from_log(lg, mel, B, T, unify, ws, den, s, d_t);
  // ---- the vocoder, one pass per run of clips: every launch stops a clip at its own length (round 5), so a run needs no common
  // padded frame count -- only the ResUNet does
  for (auto& r : runs) {
    std::vector<int> hv(3 * (size_t)r.n), hi(3 * (size_t)r.n, 0);
    for (int j = 0; j < r.n; ++j) {
      hv[j] = host[r.b0 + j];
      hv[(size_t)r.n + j] = host[B + r.b0 + j];
      hv[2 * (size_t)r.n + j] = host[2 * (size_t)B + r.b0 + j];
      hi[(size_t)r.n + j] = r.b0 + j;
    }
    launch_set_lens(d_vl, cap, hv.data(), r.n, s);
    launch_set_lens(d_gt, cap, hi.data(), r.n, s);      // (row 4 = the run's batch indices for the gather)
    launch_gather_rows(den, d_gi, vm, r.n, T, r.Tv, 128, s);
    const int64_t Llong = vocoder_out_len(h->cfg, r.Tv);
    auto plan = get_plan(h, key_of("voc_vl", r.n, r.Tv), [&](PlanBuilder& pb) {
      pb.lens_t = d_vt;
      pb.lens_tp = d_vtp;
      const BufRef peak_buf = ext(2);
      build_vocoder(pb, r.n, r.Tv, ext(0), ext(1), &peak_buf);
    }, stream);
    debug_poison(h, *plan, stream);
    RunCtx ctx{s, {vm, wlong, pk}, h->d_flags, &h->prof};
    plan->run(ctx);
    launch_peak_trim_varlen(wlong, r.n, Llong, Lmax, hop, d_vl, d_vtp, pk, wav_out + (int64_t)r.b0 * Lmax, s, h->d_flags);
  }
  VFX_API_END
}

// The spectrogram-domain twin: the per-segment body of handler_ssr_unet (eval_ssr_unet.py:77-114: sp = |STFT(wav)|, model(sp, wav))
// for a batch of clips of unequal length -- per clip the frames and reflection of its own length, the trunk's zero time padding
// behind its own last frame (unet_v2.py:103-110), the ISTFT to its own length (fDomainHelper.py:30-32); zeros past its end.
// Same requirement as vfx_restore_gsr_varlen: one padded frame count per call.
static int vfx_restore_ssr_varlen_1(vfx_handle* h, const float* wav, int B, int Lmax, const int* lengths, float* wav_out, void* stream);
int vfx_restore_ssr_varlen(vfx_handle* h, const float* wav, int B, int Lmax, const int* lengths, float* wav_out, void* stream) {
  if (!h || B <= 0 || Lmax <= 0 || !lengths) return vfx_restore_ssr_varlen_1(h, wav, B, Lmax, lengths, wav_out, stream);
  const int T = Lmax / h->cfg.hop + 1;
  const int step = std::min(kMaxVarlenClips, max_clips_per_launch(h, T, false, true, false));
  for (int b = 0; b < B; b += step) {
    const int rc = vfx_restore_ssr_varlen_1(h, wav + (int64_t)b * Lmax, std::min(step, B - b), Lmax, lengths + b,
                                            wav_out + (int64_t)b * Lmax, stream);
    if (rc) return rc;
  }
  return 0;
}
static int vfx_restore_ssr_varlen_1(vfx_handle* h, const float* wav, int B, int Lmax, const int* lengths, float* wav_out, void* stream) {
  VFX_API_BEGIN_HS(h, stream)
  VFX_CHECK(h && wav && wav_out && lengths && B > 0 && B <= kMaxVarlenClips, "bad argument");
  VFX_CHECK(h->unet[VFX_MODEL_UNET_SPEC], "vfx_restore_ssr_varlen: weights of the spectrogram ResUNet are not finalized");
  const int hop = h->cfg.hop;
  const int T = frames_of(h, Lmax), Tpad = (T + 63) / 64 * 64;
  std::vector<int> host(3 * (size_t)B);
  for (int b = 0; b < B; ++b) {
    const int Lb = lengths[b];
    VFX_CHECK(Lb > h->cfg.n_fft / 2 && Lb <= Lmax, "vfx_restore_ssr_varlen: clip %d has %d samples (need %d < length <= Lmax = %d)", b, Lb,
              h->cfg.n_fft / 2, Lmax);
    const int Tb = Lb / hop + 1;
    VFX_CHECK((Tb + 63) / 64 * 64 == Tpad, "vfx_restore_ssr_varlen: clip %d has %d frames (padded %d) but the batch's longest row pads to %d -- "
              "the clips of one call must share 64 * ceil(T / 64): bucket them by it", b, Tb, (Tb + 63) / 64 * 64, Tpad);
    host[b] = Lb;
    host[B + b] = Tb;
    host[2 * (size_t)B + b] = Tb;  // (no vocoder here)
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  int* const d_l = h->d_lens;
  int* const d_t = h->d_lens + kMaxVarlenClips;
  launch_set_lens(h->d_lens, kMaxVarlenClips, host.data(), B, s);
  const size_t nsp = (size_t)B * T * (h->cfg.n_fft / 2 + 1);
  auto plan = get_plan(h, key_of("restore_ssr_vl", B, Lmax), [&](PlanBuilder& pb) {
    auto& nm = pb.plan->named;
    const size_t o_sp = pb.alloc_f(nsp), o_cos = pb.alloc_f(nsp), o_sin = pb.alloc_f(nsp), o_re = pb.alloc_f(nsp), o_im = pb.alloc_f(nsp);
    nm["re"] = o_re;
    nm["im"] = o_im;
    vfx_handle* hh = pb.h;
    Plan* pl = pb.plan;
    pb.lens_t = d_t;
    pl->ops.push_back([=](const RunCtx& c) {  // one STFT for magnitude and phase (the reference runs two: eval_ssr_unet.py:80, unet_v2.py:96)
      launch_stft_mel(hh->fe, c.ext[0], B, Lmax, T, nullptr, reinterpret_cast<float*>(pl->bound_base + o_sp),
                      reinterpret_cast<float*>(pl->bound_base + o_cos), reinterpret_cast<float*>(pl->bound_base + o_sin), 0, hh->cfg.hop,
                      1e-8f, c.stream, d_l);
    });
    build_unet_spec(pb, B, T, arena_buf(o_sp), arena_buf(o_cos), arena_buf(o_sin), arena_buf(o_re), arena_buf(o_im));
  }, stream);
  debug_poison(h, *plan, stream);
  RunCtx ctx{s, {const_cast<float*>(wav)}, h->d_flags, &h->prof};
  plan->run(ctx);
  launch_istft(h->fe, reinterpret_cast<float*>(h->arena + plan->named["re"]), reinterpret_cast<float*>(h->arena + plan->named["im"]), B, T,
               Lmax, hop, wav_out, s, d_l);
  VFX_API_END
}

// ---------------------------------------------------------------------------------------------
// live kernel timing (bench.py roofline): HIP events around every tap-convolution launch, on the
// stream the kernels are launched on.
// ---------------------------------------------------------------------------------------------
int vfx_profile_begin(vfx_handle* h) {
  VFX_API_BEGIN_H(h)
  VFX_CHECK(h, "NULL handle");
  h->prof.enabled = true;
  h->prof.events.clear();
  h->prof.flops.clear();
  h->prof.bytes.clear();
  h->prof.design_bytes.clear();
  h->prof.bn.clear();
  VFX_API_END
}

// Synchronises, then returns: number of launches, sum of their durations (ms) and of their
// algorithmic FLOPs (2 * M * Cout * K).  Any out pointer may be NULL.
int vfx_profile_end(vfx_handle* h, int64_t* launches, double* total_ms, double* total_flops) {
  VFX_API_BEGIN_H(h)
  VFX_CHECK(h, "NULL handle");
  VFX_HIP(hipDeviceSynchronize());
  double ms = 0, fl = 0;
  FILE* dump = nullptr;
  if (const char* path = getenv("VFX_PROFILE_DUMP")) dump = fopen(path, "w");
  if (dump) fprintf(dump, "idx,kernel,M,Cout,K,nseg,ntaps0,C0,Wi,sw,ms,tflops,bytes,design_bytes\n");
  for (size_t i = 0; i < h->prof.events.size(); ++i) {
    float t = 0.f;
    VFX_HIP(hipEventElapsedTime(&t, h->prof.events[i].first, h->prof.events[i].second));
    if (dump) {
      const TapConvParams& d = h->prof.desc[i];
      int K = 0;
      for (int s2 = 0; s2 < d.nseg; ++s2) K += d.seg[s2].ntaps * d.seg[s2].C;
      char kname[64];
      if (d.nseg == 0) {  // fused ResStack layer
        snprintf(kname, sizeof(kname), "%s<%d; %d>%s", d.seg[0].ntaps == 12 ? "k_resblock_pair" : (d.seg[0].ntaps == 18 ? "k_block2d" : "k_resblock"), d.Cout,
                 d.nstages, d.hionly ? " f16" : "");
      } else if (d.up16) {  // a ConvTranspose1d of the 16-bit mode on its own kernel (upsample16.hip)
        snprintf(kname, sizeof(kname), "k_up16<%d; 128> f16", d.seg[0].C / 64);
      } else {
        bool elu = false;
        for (int s2 = 0; s2 < d.nseg; ++s2) elu = elu || d.seg[s2].act == ACT_ELU;
        snprintf(kname, sizeof(kname), "k_conv<%d; %s; %s>%s", conv_block_n(d), elu ? "true" : "false", d.split ? "true" : "false",
                 d.hionly ? " f16" : "");
      }
      fprintf(dump, "%zu,%s,%d,%d,%d,%d,%d,%d,%d,%d,%.4f,%.2f,%.0f,%.0f\n", i, kname, d.M, d.Cout, K, d.nseg, d.seg[0].ntaps, d.seg[0].C,
              d.Wi, d.sw, t, h->prof.flops[i] / (t * 1e-3) / 1e12, h->prof.bytes[i], h->prof.design_bytes[i]);
    }
    ms += t;
    fl += h->prof.flops[i];
    (void)hipEventDestroy(h->prof.events[i].first);
    (void)hipEventDestroy(h->prof.events[i].second);
  }
  if (dump) fclose(dump);
  h->prof.desc.clear();
  if (launches) *launches = (int64_t)h->prof.events.size();
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  h->prof.events.clear();
  h->prof.flops.clear();
  h->prof.bytes.clear();
  h->prof.design_bytes.clear();
  h->prof.bn.clear();
  h->prof.enabled = false;
  VFX_API_END
}

}  // extern "C"
