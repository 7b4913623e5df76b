// Internal declarations shared by the libvfx.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "vfx.h"

namespace vfx {

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
struct Error {};  // thrown after set_error(); caught at the C boundary

#define VFX_HIP(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      ::vfx::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
      throw ::vfx::Error();                                                                  \
    }                                                                                        \
  } while (0)

#define VFX_CHECK(cond, ...)         \
  do {                               \
    if (!(cond)) {                   \
      ::vfx::set_error(__VA_ARGS__); \
      throw ::vfx::Error();          \
    }                                \
  } while (0)

// Every entry point that takes a handle works on the handle's device whatever the caller's current device is, and
// leaves the caller's current device as it found it (two handles on two GPUs in one process, torch.cuda.set_device
// elsewhere).
struct DeviceGuard {
  int prev = -1, dev = -1;
  explicit DeviceGuard(int device) : dev(device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) VFX_HIP(hipSetDevice(dev));
  }
  ~DeviceGuard() {
    if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// Calls on DIFFERENT streams of one device take turns on the GPU timeline: a call first makes its stream wait for the end of the
// previous call of this library on that device and leaves an event behind for the next one.  Why (profiles/r05_two_streams.md,
// profiles/r06_two_streams_xcd.md): with two handles on two streams, k_voc_final -- plain VALU arithmetic -- running while the
// other stream's 16-bit MFMA convolutions of THIS library do (fp16 or split-bf16 operands; not the fp32 MFMA form, not torch's
// hipBLASLt kernels, not any of eight synthetic co-runners; sharing a CU is not required) delivered wrong sums in lanes 48-63 of
// single instructions, a few hundred samples per batch, off by 1e-4 .. 1e-2; every launch is correct when nothing of another
// stream runs beside it.  What is KNOWN: the condition needs this library's hand-scheduled 16-bit convolution kernels on one
// queue and a VALU-dense kernel on another; the same plans on ONE stream, or with a device-wide wait between them, are
// bit-exact.  What is NOT known: the mechanism (an open issue; the aggressors are the kernels with inline-asm loads,
// hand-counted vmcnt and LDS-DMA reads past the buffer bound, so a cause inside this library is not excluded).  Until it is, the
// library does not let its own launches overlap across streams.
//   * The wait is unconditional (an event of the stream's own past is free): no reliance on comparing stream handles, which a
//     destroyed-and-reused stream address would defeat.
//   * A call made while its stream is being CAPTURED into a hipGraph is left alone (an event of another stream cannot enter a
//     capture); a REPLAY of such a graph is not a call of this library -- callers bracket it with vfx_turn_begin / vfx_turn_end
//     (include/vfx.h; Engine.replay) or keep replays on the one stream everything else uses.
//   * Host threads: the mutex is held for the duration of a call (the next call's wait needs this call's end event, which exists
//     only once the call is enqueued), so calls on one device are ENQUEUED one at a time, plan builds included.  Handles are
//     not thread-safe anyway; multi-threaded callers serialise on this lock per device.
//   * Other PROCESSES on the same GPU are outside this guard -- and round 6 measured that a second process running this library's
//     ResUNet disturbs this process's k_voc_final just the same (every batch wrong, RAS counters silent): one process per GPU
//     (dist.py) is the supported deployment.
// VFX_NO_STREAM_TURNS=1 in the environment switches the turns off (scripts/two_streams_xcd.py needs the overlap it measures).
struct DeviceTurn {
  std::mutex mu;
  hipEvent_t done = nullptr;   // end of the last call on this device
  bool any = false;
};
DeviceTurn& device_turn(int device);  // api.cpp
bool stream_turns_enabled();          // api.cpp
// the two halves, also exported as vfx_turn_begin / vfx_turn_end (each takes the lock for its own bookkeeping only)
inline void turn_wait(DeviceTurn& d, hipStream_t s) {
  if (d.any) VFX_HIP(hipStreamWaitEvent(s, d.done, 0));
}
inline bool turn_record(DeviceTurn& d, hipStream_t s) {
  if (!d.done && hipEventCreateWithFlags(&d.done, hipEventDisableTiming) != hipSuccess) return false;
  if (hipEventRecord(d.done, s) != hipSuccess) return false;
  d.any = true;
  return true;
}
struct StreamTurn {
  DeviceTurn& d;
  std::unique_lock<std::mutex> lock;
  hipStream_t s;
  bool active = false;
  StreamTurn(int device, void* stream) : d(device_turn(device)), lock(d.mu), s(static_cast<hipStream_t>(stream)) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess) {
      (void)hipGetLastError();
      cs = hipStreamCaptureStatusNone;
    }
    active = cs == hipStreamCaptureStatusNone && stream_turns_enabled();
    if (active) turn_wait(d, s);
  }
  ~StreamTurn() {
    if (active) (void)turn_record(d, s);
  }
  StreamTurn(const StreamTurn&) = delete;
  StreamTurn& operator=(const StreamTurn&) = delete;
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device property of a kernel: done once per (kernel, device).
inline bool first_use_on_current_device(uint64_t& mask) {
  int dev = 0;
  VFX_HIP(hipGetDevice(&dev));
  const uint64_t bit = uint64_t(1) << (dev & 63);
  if (mask & bit) return false;
  mask |= bit;
  return true;
}

// ---------------------------------------------------------------------------------------------
// stage-driven tap-convolution (implicit GEMM on the MFMA) -- see conv.hip
// ---------------------------------------------------------------------------------------------
constexpr int kMaxTaps = 9;
constexpr int kMaxSegs = 3;
constexpr int kKC = 32;             // input channels per K step
constexpr int kIdentityLen = 4096;  // length of the identity scale/shift tables
constexpr int kPatchMaxRows = 192;
constexpr int kMaxVarlenClips = 1024;  // clips per launch of a varlen batch (vfx_handle::d_lens)
constexpr size_t kMaxCachedPlans = 32; // per handle (round 5: 8 -> 32 -- a file with a target alternates between more than eight (stage, B, T) keys and rebuilt plans on every call; a plan is ~0.4 MB on each side); a plan owns host + device parameter blocks (the eval handler's last
                                        // segment has a new length for every file)  // patch pixels per stage (6 row groups of 32)

enum Act { ACT_NONE = 0, ACT_LEAKY = 1, ACT_ELU = 2 };

// One K-range of the implicit GEMM: a source tensor (channels-last), its prologue
// (per-channel affine + activation, applied while the patch is staged into LDS; positions
// outside the tensor are 0 AFTER the prologue) and a tap list with K-chunked weights.
struct TapSeg {
  const float* src;    // (B, Hi, Wi, C)
  const float* scale;  // [C] or nullptr (=1)
  const float* shift;  // [C] or nullptr (=0)
  const float* wt;     // fragment-packed [C/32][ntaps][Cout/32][1024] (pack_conv)
  int C;
  int ntaps;
  int act;
  float slope;
  int src_act;         // 1: src is an ACTIVATED tensor (see TapConvParams::out_act): staged by plain copy, no prologue
  int dh[kMaxTaps];
  int dw[kMaxTaps];
};

// One stage of a launch: a 32-channel chunk of one source and the taps that read it through one
// staged patch.  Read by the kernel with scalar loads; built by build_stages() at bind time.
struct ConvStage {
  const float* src;    // source tensor + channel offset of the chunk
  const float* scale;  // prologue affine of the chunk's 32 channels
  const float* shift;
  const float* wt;     // weights of the stage's first tap, cout block 0
  int C;               // pixel stride of src in floats
  int dh0, dw0;        // patch origin relative to the tile origin
  int ntaps;
  float slope;         // LeakyReLU slope of the prologue (1 = identity, 0 = ReLU)
  int tap_stride;      // floats between the weights of consecutive taps (Cout * 32)
  int flags;           // bit 0: src is an activated tensor (LDS-DMA copy, no prologue)
  unsigned nbytes;     // bytes addressable from src (buffer descriptor bound; reads past it return 0)
  int poff[16];        // per tap (kMaxTaps used): patch row offset | column shift << 16 | row shift << 24
};
static_assert(sizeof(ConvStage) == 128, "ConvStage is read as a 128-byte record");

struct TapConvParams {
  TapSeg seg[kMaxSegs];
  int nseg;
  int total_steps;       // sum over segs of ntaps * C/32
  int B, Hi, Wi;         // input spatial extent (shared by all segments)
  int Hg, Wg;            // logical grid: one GEMM row per (b, i < Hg, j < Wg)
  int Ho, Wo, Cout;      // output tensor (B, Ho, Wo, Cout)
  int sh, sw, oh0, ow0;  // grid (i, j) -> output pixel (i*sh + oh0, j*sw + ow0), masked to Ho x Wo
  int reflect_w;         // reflect addressing along W (ReflectionPad1d) instead of zero padding
  // Folded 1-D geometry (Conv1d with dilation d viewed as a 2-D convolution with VERTICAL taps on the
  // sequence reshaped to rows of d samples): pixel (i, j) of image b is element b * img_stride + i * W + j
  // and exists iff i * W + j < limit.  0 = plain 2-D tensors (img_stride = limit = H * W); set by finish_params.
  int in_img_stride, in_limit;
  int out_img_stride, out_limit;
  int M;                 // output pixels of the launch (B * Hg * Wg, or B * out_limit when folded)
  int split;             // 1 = split-bf16 operand mode, 0 = exact fp32
  int hionly;            // 16-bit operands: split layouts, but the hi halves hold fp16 values and are the only ones
                         // loaded and multiplied (1 MFMA per product); weights packed with mode 2
  // tile / patch geometry (plan_conv): the M tile is a TH x TW block of the logical grid of one image,
  // every stage stages a PH x PW patch (P = PH * PW <= kPatchMaxRows pixels).
  int TH, TW, tw_shift;   // TH * TW <= 128, TW = 1 << tw_shift
  int tiles_h, tiles_w;   // tiles per image
  int PW, P;
  int per_tap;            // 1: the taps are too far apart for one patch -> one stage per (chunk, tap)
  int dh_min, dw_min;     // patch origin of the all-taps window
  int nstages;
  const ConvStage* stages;  // device table (bind time)
  // Phased launch (the output phases of a transposed convolution as ONE launch): Cout = nphase * cout_phase, the
  // couts [r * cout_phase, (r+1) * cout_phase) are phase r with its own stage table (stages + r * nstages) and its own
  // weight tensor.  With the output viewed as (B, T, nphase * cout_phase) this IS ConvTranspose1d's (B, T * nphase,
  // cout_phase): the input patch is fetched from HBM once for all phases.  nphase = 0 / 1: ordinary launch.
  int nphase, cout_phase;
  // Phased launch onto an ODD output width (round 4: the mel ResUNet's upsamplers, output width 2 W + 1): the two column classes
  // of one output row class cannot be viewed as channel halves of a (B, H, W', 2 C) tensor -- rows of 2 W + 1 pixels do not split
  // into pairs -- so the output is addressed in units of out_cmul = C channels: pixel index (oh * Wo + 2 j) of the TRUE tensor,
  // phase r at channel offset r * C, i.e. at true column 2 j + r; phase 1 has one column fewer (masked: ow + r < Wo).
  // 0 = ordinary addressing (pixel index * Cout).  No residual, activated output or split-K with it.
  int out_cmul;
  // Round 6: BOTH output row classes of such an upsampler in one launch -- four phases r = 2 a + b, phase r writes output row
  // 2 i + a (oh0 = 0, sh = 2) and true column 2 j + b; the four blocks of a spatial tile are neighbours in the tile order and share one
  // patch through their XCD's L2: x is read from HBM once instead of twice, one launch instead of two.  Needs out_cmul.
  int phase_rows;
  double flops_override;  // algorithmic flops when they are not 2 * M * Cout * K (phased launches)
  const float* bias;     // [Cout] or nullptr
  const float* residual; // (B, Ho, Wo, Cout) or nullptr, added in the epilogue
  // 16-bit mode, fp16 trunk (round 4): the residual as the ACTIVATED fp16 tensor fp16(LeakyReLU(r, slope)) that the launch's
  // producer chain already stores for a convolution -- the raw value is recovered as min(v, v * residual_inv_slope) (LeakyReLU
  // with a positive slope is invertible; cf. ResBlockParams::x16), so a two-launch ResStack layer keeps NO raw tensor at all.
  const float* residual_act;
  float residual_inv_slope;
  float* out;            // raw fp32 output (B, Ho, Wo, Cout) or nullptr
  // Optional ACTIVATED output for a tensor whose only consumer is the next convolution: the epilogue
  // applies that consumer's prologue (per-channel affine, LeakyReLU / ELU) once per element and stores
  // the MFMA operand form: per pixel and 32-channel chunk [32 hi bf16 | 32 lo bf16] in split-bf16 mode and 32
  // activated floats in fp32 mode (same size as fp32); in the 16-bit mode (hionly) an fp16 tensor, 2 bytes per
  // element, channels in natural order.  The consumer stages it by plain copy (LDS-DMA).
  float* out_act;
  const float* act_scale;  // [Cout] or nullptr (=1)
  const float* act_shift;  // [Cout] or nullptr (=0)
  float act_slope;         // LeakyReLU slope in [0, 1] (1 = identity)
  int act_elu;             // 1: ELU instead of LeakyReLU
  int* flags;              // the handle's sticky device flags (VFX_FLAG_F16_SATURATED); may be NULL
  // Split-K (choose_ksplit): the deep ResUNet levels have a few hundred output pixels per clip and K = 3456 .. 6912 -- a
  // launch of 100 .. 400 blocks that each walk the whole K range.  ksplit > 1: the stage table is cut into ksplit
  // consecutive ranges, block (tile, s) accumulates range s and stores its raw fp32 partial tile into slice s of `ws`
  // ([ksplit][B * out_img_stride][Cout]); k_splitk_reduce adds the slices in order and applies the epilogue (bias,
  // residual, raw / activated output).  The split depends on the per-clip geometry only, never on the batch size:
  // results do not depend on how clips are batched.
  int ksplit;
  // Set by the plan (PlanBuilder::short_clip): the clip has at most 128 padded frames (the 1-s streaming chunk).  Such a clip
  // cannot fill the chip whatever the batch is asked to be -- the deep launches are pure latency chains -- so the split aims at
  // 512 blocks per clip with slices of one stage (round 4: 3.16 -> 2.51 ms per 1-s chunk; the same rule costs a 16 x 10 s batch
  // +20 %, profiles/r04_c8_splitk_ab.txt).  A property of the CLIP, never of the batch: results stay batch-invariant.
  int short_clip;        // (< 0: a LONG clip of -short_clip x 1024 padded frames -- the split aims at 128 x that many blocks per clip)
  float* ws;
  int tuning;            // vfx_config.tuning of the handle (choose_ksplit)
  // Batches of clips of unequal length (vfx_restore_gsr_varlen; 1-D launches of the vocoder): lens[b] = length of clip b in
  // vocoder frames (T_b + T_b % 2 + 4), a device array owned by the handle and refilled by every call; the clip's input /
  // output sequence ends at lens[b] * lens_mul_in / lens_mul_out positions (the launch's own units: a phased upsampler
  // addresses its output in INPUT positions).  NULL = every clip has the full length (all other launches).  No split-K with it.
  const int* lens;
  int lens_mul_in, lens_mul_out;
  // Set by PlanBuilder::add_conv_phased: this phased launch (a ConvTranspose1d of the vocoder's 16-bit mode: activated fp16 source,
  // one fp16 output) runs on k_up16 (upsample16.hip) -- one block per tile of input positions and ALL its output phases, the patch
  // staged once -- instead of k_conv's block per (tile, phase, cout range).  Same stage tables, same sums.
  int up16;
};

// One fused TFGAN ResStack layer (resblock.hip): y = x + conv2(LeakyReLU(conv1(LeakyReLU(x)) + b1)) + b2.
struct ResBlockParams {
  const float* x;   // (B, T, C) raw fp32 input, also the residual
  float* y;         // (B, T, C) raw fp32 output
  const float* w1;  // conv1 (k3, dilation dil), fragment-packed [C/32][3][C/32][1024] (pack_conv)
  const float* w2;  // conv2 (k3, dilation 1)
  const float* b1;  // [C]
  const float* b2;
  float slope;      // LeakyReLU slope of both activations
  int B, T, C, dil;
  // tile geometry (plan_resblock): h tile = TH x W1 positions, position(li, lj) = base + li * rowstride + lj
  int fold;         // 1: the sequence is viewed as rows of `dil` samples (conv1's taps are vertical)
  int TH, W1, TWo;  // TWo = outputs per tile row (W1 - 2)
  int tiles_h, tiles_w;
  int PW, P;        // x patch: PH x PW pixels, P <= kPatchMaxRows
  int poff[3];      // patch row offset of conv1's taps
  int hionly;       // fp16 operands in the hi halves only, cf. TapConvParams::hionly
  int* flags;       // the handle's sticky device flags (VFX_FLAG_F16_SATURATED); may be NULL
  // Wide stacks of the 16-bit mode (resblock_w64.hip, C = 256): conv1 reads the activated fp16 tensor
  // xa = fp16(LeakyReLU(x)) (2 bytes per element) through the LDS-DMA engine, no arithmetic on it; besides y the kernel
  // writes ya = fp16(LeakyReLU(y, act_slope)) for the next consumer (NULL: not needed).  Weights: pack_conv mode 3.
  //   x16 == 0 (VFX_TUNE_F32_TRUNK): two-form trunk -- x is the raw fp32 residual, y the raw fp32 output (12 bytes per element);
  //   x16 == 1: the activated fp16 tensor is the ONLY form of the trunk (4 bytes per element): x and y are NULL, the residual
  //   is recovered from xa (LeakyReLU with a positive slope is invertible: x = xa >= 0 ? xa : xa / slope -- the same relative
  //   precision as fp16(x)), ya must be there.
  int asrc;
  // fp16 residual trunk of the 16-bit mode (the default; VFX_TUNE_F32_TRUNK = 0 here): x and y point to fp16 tensors (B, T, C),
  // 2 bytes per element, channels in natural order.  The layer's arithmetic is unchanged -- fp16 operands, fp32 accumulation,
  // the residual added in fp32 in registers -- only what travels between two launches is rounded (saturation is flagged).
  // y may be NULL when only ya is consumed (the last layer of a stack in front of an upsampler).
  int x16;
  int tile_m;       // h positions per tile: 0 = 128 (k_resblock, resblock_w64, resblock_r128); resblock_rw: 128 or 256
  // Layer pair (resblock_rw.hip, PAIR): dil2 > 0 = a second layer (dilation dil2, weights w1b .. b2b, same slope) follows in the
  // same launch; y is ITS output, the first layer's output is never stored.
  int dil2;
  const float* w1b;
  const float* w2b;
  const float* b1b;
  const float* b2b;
  int r128;         // set by plan_resblock: resblock_r128.hip runs this layer (16-bit mode, C = 128)
  int rw;           // set by plan_resblock: the persistent register-weights kernel runs this layer (resblock_rw.hip: 16-bit mode, C = 64)
  const float* xa;
  float* ya;
  float act_slope;
  // 2-D ConvBlockRes mode (plan_block2d): x, y are (B, H, W, C), both convolutions 3x3, folded BatchNorm affines
  int geo2d, H, W;
  const float* sc1;  // bn1 scale / shift [C]: prologue of conv1
  const float* sh1;
  const float* sc2;  // bn2 scale / shift [C]: prologue of conv2, applied to h
  const float* sh2;
  // entry block (k_resblock<.., IN1>): x is the single-channel input plane (B, H, W), w1 the [tap][32] conv1 weights, bn1 the
  // scalar affine below, wsc / bsc the 1x1 shortcut (weight and bias per output channel); sc1 / sh1 are unused
  int in1;
  float in1_scale, in1_shift;
  const float* wsc;
  const float* bsc;
  // two-source block (k_resblock<.., SC2>; decoder level 1): the input is cat(x, x2) (C channels each), w1 / w1x2 the conv1 weights
  // of the two sources, sc1 / sh1 hold 2 C channels, wsc / wsc2 the fragment-packed 1x1 shortcut of each source, bsc its bias
  int two_src;
  const float* x2;
  const float* w1x2;
  const float* wsc2;
  int poff9[9];      // conv1: patch row offset of tap (dy, dx)
  int hoff9[9];      // conv2: h row offset of tap (dy, dx)
  // set by plan_resblock: multiply-shift reciprocals, n / d = (n * inv) >> 20 (exact for n < 512, d <= 320), and the tile ->
  // (image, tile row, tile column) split as (tile * inv_tpi) >> 32 etc. (exact for tile < 2^32 / d)
  unsigned inv_pw, inv_w1;
  // ceil(2^32 / n) for n = tiles_w, tiles_w * tiles_h: 33 bits when n = 1 (one tile per image: short sequences) -- div_recip()
  unsigned long long inv_tiles_w, inv_tiles_per_img;
  int recip_ok;      // plan_block2d: the reciprocal tile split is exact for this launch's tile count
  int patch_rows;    // set by plan_resblock: rows of a patch buffer when they are not tile_m + 64 (resblock_w64.hip: 160)
  int tuning;        // vfx_config.tuning of the handle: which kernel family runs the layer (plan_resblock)
  // Batches of clips of unequal length (cf. TapConvParams::lens): clip b's sequence ends at lens[b] * lens_mul positions; T stays
  // the stride between clips.  NULL = T for every clip.  1-D layers only.
  const int* lens;
  int lens_mul;
  // Timing builds only (-DVFX_TIMING, scripts/phase_timing.py): [tile][wave][16] s_memtime stamps of the 4-wave kernels' phases
  unsigned long long* timing;
};
bool resblock_supported(int C);
int resblock_block_waves(const ResBlockParams& hp);
// resblock_rw.hip: C = 64, 16-bit mode -- persistent blocks, weights in registers, next patch prefetched into registers
// resblock_w64.hip: the wide layer (C = 256, 16-bit mode) as 4-wave blocks of 64-cout waves, two blocks per CU
bool resblock_w64_supported(int C);
// resblock_r128.hip: C = 128, 16-bit mode -- 4-wave blocks, two per CU, x read once (the residual stays in registers)
bool resblock_r128_pair_ok(int C, int dil, int dil2, int tuning);  // two layers (dil, dil2) as one launch: (1, 3)
int resblock_r128_patch_rows();
void launch_resblock_r128(const ResBlockParams& hp, const ResBlockParams* dparams, hipStream_t stream);
int resblock_w64_patch_rows();
void launch_resblock_w64(const ResBlockParams& hp, const ResBlockParams* dparams, hipStream_t stream);
int resblock_rw_tile(int tuning);
bool resblock_rw_pair_ok(int C, int dil, int dil2, int tuning);  // this pair of consecutive layers can run as one launch
void launch_resblock_rw(const ResBlockParams& hp, const ResBlockParams* dparams, hipStream_t stream);
// compute units of the current device (cached per device): grid size of the persistent kernels
int cu_count_of_current_device();
bool block2d_supported(int C);
void plan_resblock(ResBlockParams& p);
void plan_block2d(ResBlockParams& p);
void launch_resblock(const ResBlockParams& hp, const ResBlockParams* dparams, hipStream_t stream);
double resblock_flops(const ResBlockParams& hp);

// Geometry + taps of a Conv1d (kernel K, dilation dil, 'same' padding) over (B, T, C) tensors as seg[0] of p:
// plain 1-D when the taps fit one patch, folded (rows of `dil` samples, vertical taps) otherwise.
void set_conv1d_geometry(TapConvParams& p, int B, int T, int K, int dil, bool reflect);
void finish_params(TapConvParams& p);  // fills total_steps, M and the tile / patch geometry, validates
int count_stages(const TapConvParams& p);
int choose_ksplit(const TapConvParams& p);  // after finish_params
void launch_splitk_reduce(const TapConvParams& hp, hipStream_t stream);  // hp: absolute pointers
void build_stages(const TapConvParams& p, const float* ones, const float* zeros, ConvStage* out);  // p: absolute pointers
void launch_conv(const TapConvParams& hp, const TapConvParams* dparams, hipStream_t stream);
int conv_block_n(const TapConvParams& hp);
bool block2d32_ok(const ResBlockParams& hp);  // block2d32.hip
bool block2d32_has_tile(int TH, int W1);
void launch_block2d32(const ResBlockParams& hp, const ResBlockParams* dparams, hipStream_t stream);
bool upsample16_ok(const TapConvParams& hp);  // upsample16.hip
void launch_upsample16(const TapConvParams& hp, const TapConvParams* dparams, hipStream_t stream);
double conv_flops(const TapConvParams& hp);

// ---------------------------------------------------------------------------------------------
// front-end tables + small kernels -- see stft.hip / small_ops.hip
// ---------------------------------------------------------------------------------------------
constexpr int kMelNnzMax = 2048;  // non-zeros of the mel filterbank that k_stft_mel keeps in LDS (2018 for the 128-band HTK table)

struct FrontEndTables {
  float* window = nullptr;    // [2048] periodic Hann
  float* twiddle = nullptr;   // [1024] float2 e^{-2 pi i m / 1024}
  float* rtwiddle = nullptr;  // [1025] float2 e^{-2 pi i k / 2048}
  float* fb_val = nullptr;    // packed non-zeros of the mel filterbank, band-major
  int* fb_start = nullptr;    // [128] first frequency bin of band m
  int* fb_off = nullptr;      // [129] offsets into fb_val
  int fb_nnz = 0;             // entries of fb_val (<= kMelNnzMax: k_stft_mel keeps them in LDS)
  float* voc_inv_weight = nullptr;  // [128] 1 / mel band weight
};

// lens != nullptr (device, [B]): clip b holds lens[b] <= L samples in its row of L (batches of clips of unequal length)
void launch_stft_mel(const FrontEndTables& t, const float* wav, int B, int L, int T, float* mel, float* sp,
                     float* cosp, float* sinp, int log10_mel, int hop, float eps, hipStream_t stream,
                     const int* lens = nullptr);
void launch_mel_project(const FrontEndTables& t, const float* sp, int64_t rows, float* mel, hipStream_t stream);
void launch_istft(const FrontEndTables& t, const float* re, const float* im, int B, int T, int L, int hop, float* wav,
                  hipStream_t stream, const int* lens = nullptr);  // lens: samples per clip of a varlen batch (device, [B])

void launch_prep_logmel(const float* mel_linear, int B, int T, int Tpad, float* x, int* flags, hipStream_t s,
                        const int* lens_t = nullptr);  // lens_t: frames per clip of a varlen batch (device, [B])
void launch_prep_spec(const float* sp, int B, int T, int Tpad, float* x, hipStream_t s, const int* lens_t = nullptr);
void launch_conv_c1(const float* x, int B, int H, int W, const float* w9x32, float scale, float shift, float slope,
                    const float* wsc32, const float* bsc32, float* h, float* sc, hipStream_t s);
void launch_avgpool2(const float* x, int B, int H, int W, int C, float* y, hipStream_t s);
void launch_final_1x1(const float* y, int B, int Tpad, int W, const float* w32, float bias, int mode, int T,
                      const float* aux0, const float* aux1, float* out0, float* out1, hipStream_t s);
void launch_voc_prep(const float* mel, int B, int T, int Tp, const float* inv_weight, float amp_floor, float min_db,
                     float range, float* cond, hipStream_t s, const int* lens_t = nullptr);
// x_f16: x is an fp16 tensor (the fp16 trunk of the 16-bit mode); lens: vocoder frames per clip of a varlen batch, hop samples each
void launch_voc_final(const float* x, int x_f16, int B, int T, int C, const float* w, float bias, float slope, float* wav,
                      unsigned* peak, hipStream_t s, const int* lens = nullptr, int hop = 1);
// d_lens[r * cap + b] = host[r * B + b], r < 3 (samples, frames, vocoder frames per clip): as kernel arguments, in stream order
void launch_set_lens(int* d_lens, int cap, const int* host, int B, hipStream_t s);
void launch_copy_rows_masked(const float* src, float* dst, int B, int T, int F, const int* lens_t, hipStream_t s);
// dst (n, Tg, F) compact <- rows of src (B, T, F) of the clips idx[j] (rows >= T: zeros);  ... and back (rows < min(T, Tg))
void launch_gather_rows(const float* src, const int* idx, float* dst, int n, int T, int Tg, int F, hipStream_t s);
void launch_scatter_rows(const float* src, const int* idx, float* dst, int n, int T, int Tg, int F, hipStream_t s);
void launch_peak_trim_varlen(const float* wav_long, int B, int64_t Llong, int L, int hop, const int* lens_l, const int* lens_tp,
                             const float* peak, float* out, hipStream_t s, int* flags);
void launch_from_log(const float* logmel, const float* mel_in, int B, int T, int unify, float* sums, float* mel_out,
                     hipStream_t s, const int* lens_t = nullptr);
void launch_peak_trim(const float* wav_long, int B, int64_t Llong, int L, float* ws, bool have_peak, float* out,
                      hipStream_t s, int* flags = nullptr);  // flags: VFX_FLAG_PEAK_NORMALISED when a clip was divided by its peak
void launch_spectral_metrics(const float* est, const float* tgt, int B, int T, int F, double* ws, float* out, hipStream_t s);
int64_t count_nonfinite(const float* p, int64_t n, hipStream_t s);  // debug aid, synchronises
void launch_chunk_gather(const float* x, int B, int L, int win, int hop, int lead, int n_chunks, float* chunks,
                         hipStream_t s);
void launch_chunk_ola(const float* frames, const float* window, float scale, int B, int n_chunks, int win, int hop,
                      int lead, int L, float* y, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// handle-side data structures
// ---------------------------------------------------------------------------------------------
struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
};

// Long-lived device allocations (tables, weights, plan parameter blocks).
struct DeviceBlob {
  std::vector<void*> allocs;
  void* alloc(size_t bytes);
  float* upload(const float* p, size_t n);
  float* upload(const std::vector<float>& v) { return upload(v.data(), v.size()); }
  int* upload_i(const std::vector<int>& v);
  void release();
  ~DeviceBlob() { release(); }
};

// Offsets into the workspace arena, assigned at plan-build time by a first-fit free list.
struct ArenaPlanner {
  struct Block { size_t off, size; bool free; };
  std::vector<Block> blocks;
  size_t high = 0;
  size_t alloc(size_t bytes);
  void free(size_t off);
};

// A buffer is either a slice of the arena (resolved when the plan is bound to an arena base)
// or one of the caller's tensors (resolved per call).
struct ConvProfile {  // HIP-event timing of every convolution launch (vfx_profile_*)
  bool enabled = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  std::vector<double> flops;
  std::vector<double> bytes;  // algorithmic HBM bytes of the launch
  std::vector<double> design_bytes;  // what the kernel's data layout moves (two-form trunks carry their fp16 copies)
  std::vector<int> bn;
  std::vector<TapConvParams> desc;  // fused ResStack layers are described as M = B*T, Cout = C, nseg = 0
};

struct RunCtx {
  hipStream_t stream;
  float* ext[8];
  int* flags;
  ConvProfile* prof = nullptr;
};

struct Plan {
  std::vector<std::function<void(const RunCtx&)>> ops;
  std::vector<TapConvParams> host_params;  // arena-relative until bind()
  std::vector<TapConvParams> abs_params;   // the same with absolute pointers, as uploaded by bind()
  TapConvParams* dev_params = nullptr;
  ConvStage* dev_stages = nullptr;
  std::map<size_t, std::vector<TapSeg>> phase_segs;  // phased launches: host_params index -> per-phase segment
  std::vector<ResBlockParams> host_rb;     // arena-relative until bind()
  ResBlockParams* dev_rb = nullptr;
  DeviceBlob blob;
  size_t arena_bytes = 0;
  char* bound_base = nullptr;
  double conv_flops = 0;
  int n_conv = 0;
  std::map<std::string, size_t> named;  // named arena offsets (bytes) of stage-level buffers
  uint64_t last_use = 0;                // handle tick of the last call (LRU eviction, get_plan)
  bool pinned = false;                  // a hipGraph was captured from this plan: its parameter blocks must outlive the graph
  void run(const RunCtx& ctx);  // api.cpp (debug hooks: VFX_POISON_ARENA=2, VFX_DEBUG_NAN)
};

// Arena-relative pointer encoding used inside TapConvParams until bind_plan(): offset + 1
// (so that offset 0 is distinguishable from nullptr).
inline const float* rel_ptr(size_t off) { return reinterpret_cast<const float*>(off + 1); }

struct PlanBuilder {
  vfx_handle* h;
  Plan* plan;
  ArenaPlanner arena;
  int short_clip = 0;  // copied into every TapConvParams added from here on (TapConvParams::short_clip)
  // Batches of clips of unequal length (vfx_restore_gsr_varlen): device arrays [B] of the handle, refilled by every call --
  // frames per clip (T_b = L_b / hop + 1) and vocoder frames per clip (T_b + T_b % 2 + 4).  NULL: every clip has the plan's T.
  const int* lens_t = nullptr;
  const int* lens_tp = nullptr;
  // set by build_vocoder: its launches never split K (split-K is the deep ResUNet levels' tool; a vocoder launch of a short clip
  // would otherwise sum in another order than the same clip inside a varlen batch, whose launches cannot split -- with this
  // every arithmetic mode gives a clip the same bits in both entry points)
  bool no_splitk = false;
  // returns arena offset in BYTES
  size_t alloc_f(int64_t nfloat) { return arena.alloc((size_t)nfloat * sizeof(float)); }
  void free(size_t off) { arena.free(off); }
  // Adds a tap-convolution whose src/residual/out pointers are arena offsets encoded as
  // (const float*)offset; they are rebased in Plan::bind.
  void add_conv(TapConvParams p);
  // p.seg[0] carries the UNION of the phases' taps (geometry only); phases[r] = source / prologue / taps / weights of phase r
  void add_conv_phased(TapConvParams p, const std::vector<TapSeg>& phases);
  void add_resblock(ResBlockParams p);  // x / y are arena offsets encoded with rel_ptr()
};

struct ConvBlockW {
  int cin = 0, cout = 0, nsrc = 1;  // nsrc = 2: the input is cat(src0, src1), cin/2 channels each
  float *bn1_scale = nullptr, *bn1_shift = nullptr, *bn2_scale = nullptr, *bn2_shift = nullptr;
  float* w1[2] = {nullptr, nullptr};   // conv1 packed per source
  float* w2 = nullptr;                 // conv2 packed
  float* wsc[2] = {nullptr, nullptr};  // 1x1 shortcut packed per source (cin != cout)
  float* bsc = nullptr;
  bool shortcut = false;
};

struct DecoderW {
  int cin = 0, cout = 0;
  float *bn_scale = nullptr, *bn_shift = nullptr;
  float* wT[4] = {nullptr, nullptr, nullptr, nullptr};  // packed per output parity class (a*2+b)
  ConvBlockW blocks[4];
};

struct UNetWeights {
  float c1_scale = 1.f, c1_shift = 0.f;
  float *c1_w = nullptr, *c1_wsc = nullptr, *c1_bsc = nullptr;
  ConvBlockW enc[6][4];
  ConvBlockW bott;
  DecoderW dec[6];
  ConvBlockW after;
  float* final_w = nullptr;
  float final_b = 0.f;
};

struct VocConvW {
  float* w = nullptr;     // packed (per phase for transposed convs: w_phase[r])
  float* bias = nullptr;  // [cout]; transposed convs: repeated once per output phase ([stride][cout])
  std::vector<float*> w_phase;
  int cin = 0, cout = 0;
  int mode = 0;  // pack_conv mode the weights were packed with (follows the form of the source tensor in the plan)
};

struct VocoderWeights {
  std::vector<VocConvW> cond;   // k3 convs
  VocConvW pre;                 // k7 reflect
  std::vector<VocConvW> up;     // transposed convs
  std::vector<std::vector<std::pair<VocConvW, VocConvW>>> res;
  float* final_w = nullptr;     // [7][C]
  float final_b = 0.f;
  int final_c = 0;
  bool needs_strict = false;    // 16-bit mode: a weight tensor does not fit fp16 operands (f16_weight_issue)
};

// Conv weights -> MFMA fragment order [C/chunk][ntaps][Cout/32][1024 floats] (conv.hip); mode: 0 fp32, 1 split-bf16
// (hi | lo), 2 fp16 in the hi fragments (32-channel chunks, raw sources of the 16-bit mode), 3 fp16 with 64-channel
// chunks (activated fp16 sources of the 16-bit mode); conv_chunk(mode) = input channels per chunk.
int conv_chunk(int mode);
int stage_channels(const TapConvParams& p, const TapSeg& S);
std::vector<float> pack_conv(const float* w, int Cout, int CinTotal, int KH, int KW, int c_lo, int C,
                             const std::vector<std::pair<int, int>>& taps, int mode);
std::vector<float> pack_conv_transposed(const float* w, int Cin, int Cout, int KH, int KW,
                                        const std::vector<std::pair<int, int>>& taps, int mode);
void rows_to_fragments(std::vector<float>& packed, int Cout, int mode);  // [..][Cout][32] rows -> fragment order
// Set by rows_to_fragments when a tensor packed as fp16 operands (modes 2, 3) does not fit fp16 (|w| > 65504, or the whole
// tensor deep in the subnormal range); the weight builders reset it before and read it after packing a model.
bool& f16_weight_issue();
void launch_or_flags(int* flags, int bits, hipStream_t s);  // small_ops.hip: flags |= bits, in stream order

}  // namespace vfx

struct vfx_handle {
  int device = 0;
  vfx_config cfg{};
  std::map<std::string, vfx::HostTensor> staged[4];
  vfx::FrontEndTables fe;
  vfx::DeviceBlob blob;  // front-end tables + weights
  std::shared_ptr<vfx::UNetWeights> unet[2];
  std::shared_ptr<vfx::VocoderWeights> voc;
  std::map<std::string, std::shared_ptr<vfx::Plan>> plans;  // at most kMaxCachedPlans, least recently used evicted
  std::vector<std::shared_ptr<vfx::Plan>> retired;  // evicted plans whose device blocks are freed in batches (get_plan)
  uint64_t plan_tick = 0;
  char* arena = nullptr;
  size_t arena_bytes = 0;
  int* d_flags = nullptr;
  int* d_lens = nullptr;     // [6][kMaxVarlenClips]: samples / frames / vocoder frames per clip of the varlen call in flight; frames /
                             // batch index of the clips of its ResUNet group in flight (row 5 unused)
  char* scratch = nullptr;   // the tensors between the plans of a varlen call (api.cpp: ensure_scratch), grow-only
  size_t scratch_bytes = 0;
  float* d_ones = nullptr;   // identity prologue tables (kIdentityLen floats)
  float* d_zeros = nullptr;
  vfx::ConvProfile prof;
};

namespace vfx {
void init_front_end(vfx_handle* h);
void set_mel_filterbank(vfx_handle* h, const float* fb /*1025x128*/);
void bind_plan(vfx_handle* h, Plan& plan);  // ensures the arena is large enough and rebases the plan on it
std::shared_ptr<UNetWeights> build_unet_weights(vfx_handle* h, int model);
std::shared_ptr<VocoderWeights> build_vocoder_weights(vfx_handle* h);

// plan builders: append the ops of one stage to `pb`.  Buffers named *_off are arena byte
// offsets; ext slots index RunCtx::ext.
//   unet: input = ext[in_slot] or arena (in_off); mel model writes (B,T,128) log-mel to ext[out_slot] / arena.
struct BufRef {
  bool ext = false;
  int slot = 0;
  size_t off = 0;
};
void build_unet_mel(PlanBuilder& pb, int B, int T, BufRef mel_linear, BufRef logmel_out);
void build_unet_spec(PlanBuilder& pb, int B, int T, BufRef sp, BufRef cosb, BufRef sinb, BufRef re_out, BufRef im_out);
void build_vocoder(PlanBuilder& pb, int B, int T, BufRef mel_linear, BufRef wav_out, const BufRef* peak = nullptr);
int64_t vocoder_out_len(const vfx_config& cfg, int T);
}  // namespace vfx
