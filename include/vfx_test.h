/*
 * vfx_test.h -- C ABI of libvfx_test.so: kernel-level entry points for the parity tests.
 *
 * NOT part of the drop-in boundary (include/vfx.h, libvfx.so): nothing here corresponds to a call the reference makes.  Each
 * entry point runs ONE kernel family of libvfx.so in isolation -- weights handed over in PyTorch layout on the host, packed on
 * the fly, the launch synchronised -- so that tests/test_gpu_kernels.py can compare it with a float64 torch expression of the
 * same operator.  libvfx_test.so is linked against libvfx.so and uses its internals; the product (voicefixer_main_amd/models.py,
 * handlers.py, dist.py, bench.py's timed path) never loads it.
 */
#ifndef VFX_TEST_H_
#define VFX_TEST_H_

#include "vfx.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * vfx_op_conv: generic tap-convolution on channels-last activations.
 *   x (B, H, W, Cin) -> y (B, H, W, Cout); weight in PyTorch Conv2d layout (Cout, Cin, kh, kw)
 *   on the HOST; scale/shift (Cin) HOST arrays or NULL (identity prologue);
 *   act: 0 none, 1 leaky(slope), 2 elu; bias (Cout) HOST or NULL; residual device or NULL;
 *   dil_w: dilation along W; reflect_w: reflect padding along W instead of zeros.
 */
int vfx_op_conv(vfx_handle* h, const float* x, int B, int H, int W, int Cin, const float* weight,
                int Cout, int kh, int kw, int dil_w, int reflect_w, const float* scale,
                const float* shift, int act, float slope, const float* bias,
                const float* residual, float* y, void* stream);
/* ConvTranspose (stride s, PyTorch layout (Cin, Cout, kh, kw) on the HOST):
 *   2-D: kh=kw=3, s=2, padding 0, output pruned to (2H, 2W+1) or (2H, 2W) when prune_w; without a bias and for H, W >= 2 it runs
 *        the ResUNet plan's form (two phased launches, one per output row class), else four parity launches;
 *   1-D: kh=1, kw=2s, padding s/2+s%2, output_padding s%2, output (B,1,W*s,Cout). */
int vfx_op_conv_transpose(vfx_handle* h, const float* x, int B, int H, int W, int Cin,
                          const float* weight, int Cout, int kh, int kw, int stride, int prune_w,
                          const float* scale, const float* shift, int act, float slope,
                          const float* bias, float* y, void* stream);

/* One TFGAN ResStack layer (vocoder layer table, oracle/vocoder.py) on channels-last (B, T, C) tensors:
 *   y = x + conv2(LeakyReLU(conv1(LeakyReLU(x)) + b1)) + b2,  conv1: k3 with dilation `dil`, conv2: k3.
 * w1 / w2 in PyTorch Conv1d layout (C, C, 3), b1 / b2 (C), all on the HOST.  fused != 0 runs the single-launch
 * kernel (C = 64 or 128, precision 1); fused == 0 the two-launch form with the activated intermediate tensor. */
int vfx_op_resblock(vfx_handle* h, const float* x, int B, int T, int C, const float* w1, const float* b1,
                    const float* w2, const float* b2, int dil, float slope, int fused, float* y, void* stream);

/* Host-only (no GPU, no handle): the tile geometry the plan gives one fused ResStack layer -- or a layer pair, dil2 > 0 -- of
 * C channels over sequences of T positions in precision mode `precision`.  out[12] = fold, TH, W1, TWo, tiles_h, tiles_w, PW, P,
 * tile_m, rw, patch_rows, asrc (ResBlockParams; the trunk form is the one the vocoder plan would use: fp16 unless VFX_TUNE_F32_TRUNK).  The CPU tests use it to check that the tiles cover every position exactly once. */
int vfx_plan_resblock_geometry(int C, int T, int dil, int dil2, int precision, int* out);
/* ... for a handle configured with vfx_config.tuning = `tuning` (the function above is tuning = 0). */
int vfx_plan_resblock_geometry_tuned(int C, int T, int dil, int dil2, int precision, int tuning, int* out);
/* Host-only: tile and patch geometry plan_conv gives a tap convolution over an (Hg, Wg) output grid with `ntaps` taps at offsets
 * (dh[t], dw[t]).  out[6] = TH, TW, PW, P, per_tap, tiles_h * tiles_w. */
int vfx_plan_conv_geometry(int Hg, int Wg, int ntaps, const int* dh, const int* dw, int* out);
/* Host-only: the tile geometry of a fused 2-D ConvBlockRes of the ResUNets (plan_block2d) over (H, W) images of C channels;
 * kind 0 = identity block, 1 = entry block (Cin = 1), 2 = two-source block.  out[8] = TH, W1, TWo, tiles_h, tiles_w, PW, P,
 * tile_m (0 = 128 positions).  Returns 1 where the plan refuses (e.g. kind != 0 under VFX_TUNE_SMALL_2D_TILES). */
int vfx_plan_block2d_geometry(int C, int H, int W, int kind, int tuning, int* out);

/* Two consecutive ResStack layers (dilations dil, dil2) as ONE launch: y = layer_b(layer_a(x)), the intermediate tensor never
 * leaves the CU.  precision 2 only; C = 64 (resblock_rw.hip): dil <= 32, dil2 <= 62; C = 128 (resblock_r128.hip): dil <= 16,
 * dil2 <= 4 -- what the vocoder plan pairs: dilations (1, 3) and, at C = 64, (9, 27) of the 44.1 kHz stack.  Weights / biases as in vfx_op_resblock, on the HOST. */
int vfx_op_resblock_pair(vfx_handle* h, const float* x, int B, int T, int C, const float* wa1, const float* ba1,
                         const float* wa2, const float* ba2, int dil, const float* wb1, const float* bb1,
                         const float* wb2, const float* bb2, int dil2, float slope, float* y, void* stream);

/* One fused 2-D ConvBlockRes of the ResUNets (models/components/modules.py:223-271; Cin == Cout = C in {32, 64},
 * identity shortcut): y = x + conv2(lrelu(bn2(conv1(lrelu(bn1(x)))))) with 3x3 convolutions in ONE launch (h stays
 * in LDS).  x, y (B, H, W, C) on the device; w1, w2 (C, C, 3, 3) and the folded eval-mode BatchNorm affines
 * sc1, sh1, sc2, sh2 [C] on the HOST.  precision 1 only. */
int vfx_op_block2d(vfx_handle* h, const float* x, int B, int H, int W, int C, const float* w1,
                   const float* sc1, const float* sh1, const float* w2, const float* sc2,
                   const float* sh2, float slope, float* y, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VFX_TEST_H_ */
