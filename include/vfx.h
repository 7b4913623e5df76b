/*
 * vfx.h -- C ABI of libvfx.so: MI355X-native (gfx950) VoiceFixer 44.1 kHz inference hot path.
 *
 * Drop-in boundary for the reference haoheliu/voicefixer_main.  The reference is pure
 * Python/PyTorch and has no FFI of its own; each entry point below replaces the arithmetic
 * of one reference call (file:line cited) and is what a ctypes binding inside the
 * reference's own modules would bind.
 *
 * Conventions
 *   - all tensor pointers are DEVICE pointers to contiguous fp32 unless stated otherwise;
 *     the caller (PyTorch) owns inputs and outputs, the library never frees them;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     all work is enqueued on it, no entry point synchronises the device except
 *     vfx_create / vfx_load_tensor / vfx_finalize_weights / vfx_reserve / vfx_take_flags;
 *   - the handle owns a copy of the weights and one workspace arena; it is NOT thread-safe;
 *   - calls on DIFFERENT streams of one device (two handles, or one handle moved between streams) take turns on the GPU
 *     timeline: the library does not let its own launches of two streams overlap (profiles/r05_two_streams.md); a call made
 *     while its stream is being captured into a hipGraph is exempt -- bracket the graph's replays with vfx_turn_begin /
 *     vfx_turn_end when another stream of the device may run calls of this library at the same time;
 *   - every function returns 0 on success, non-zero on failure; vfx_last_error() then
 *     returns a human-readable message (thread-local).
 *   - T = L / hop + 1 frames (center=True framing); Tpad = 64*ceil(T/64).
 */
#ifndef VFX_H_
#define VFX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vfx_handle vfx_handle;

#define VFX_MAX_STAGES 8

/* model ids for weights / plans */
enum { VFX_MODEL_UNET_MEL = 0, VFX_MODEL_UNET_SPEC = 1, VFX_MODEL_VOCODER = 2,
       VFX_MODEL_FRONTEND = 3 /* buffers of f_helper / mel: "mel.fb" (1025,128) */ };

/* sticky device-side flags returned by vfx_take_flags */
enum {
  VFX_FLAG_NEGATIVE_INPUT = 1, /* to_log saw a negative value (pytorch_util.py:158) */
  VFX_FLAG_F16_SATURATED = 2,  /* precision 2: an activation of the vocoder left the fp16 range (|x| > 65504, or was
                                  not a number) and was clamped -- the result of that call is not trustworthy; re-run
                                  it with precision 1 */
  VFX_FLAG_PEAK_NORMALISED = 4 /* vfx_restore_gsr divided a clip by its peak (> 1): the reference prints "Warning:
                                  Exceed energy limit" there (eval_gsr_voicefixer.py:68-70); informational */
};

/* vfx_config.tuning: kernel-selection switches for A/B measurements and bisecting (0 = the shipped configuration; every
 * bit selects an older / simpler form of the same arithmetic, results stay within the mode's tolerances).  Read when a
 * plan is built; vfx_create reports a non-zero mask on stderr.  Every bit is exercised by the GPU test suite. */
enum {
  VFX_TUNE_NO_FUSED_STACKS = 1,    /* vocoder ResStack layers (C = 64, 128) as two tap-convolution launches per layer */
  VFX_TUNE_NO_FUSED_WIDE = 2,      /* ... the C = 256 layers of the 16-bit mode as two launches per layer (two-form trunk) */
  VFX_TUNE_NO_FUSED_UNET = 4,      /* ConvBlockRes of the ResUNets as two launches each (fused: the identity-shortcut blocks at
                                      C = 32, 64, the entry block and the two-source block of the full-resolution level) */
  VFX_TUNE_NO_PERSISTENT_C64 = 8,  /* 16-bit mode, C = 64 layers on k_resblock instead of the persistent kernel */
  VFX_TUNE_NO_PAIRS = 16,          /* 16-bit mode, C = 64 / 128: one launch per layer (no layer pairs) */
  VFX_TUNE_NO_SPLITK = 32,         /* no split-K in the deep ResUNet levels */
  VFX_TUNE_F32_TRUNK = 64,         /* 16-bit mode: the residual trunk of the fused ResStacks (C = 64 / 128 / 256) travels as fp32
                                      between the layers (the round-3 form: 8 - 12 bytes per element and layer) instead of
                                      fp16 (4 bytes per element and layer; the sums themselves are fp32 in registers either way) */
  VFX_TUNE_SMALL_2D_TILES = 128,   /* fused ConvBlockRes of the ResUNets at C = 32: 8 x 16 / 16 x 8 h tiles (84 outputs per 128 positions)
                                      instead of 16 x 16 (196 per 256); the entry block (Cin = 1) and the two-source block of that
                                      level, which exist on 16 x 16 tiles only, as two launches each */
  VFX_TUNE_NO_FUSED_UPSAMPLERS = 512, /* 16-bit mode: the vocoder's ConvTranspose1d upsamplers as phased tap-convolution launches (one block per
                                      tile, phase and cout range) instead of k_up16 (one block per tile of input positions, all phases) */
  VFX_TUNE_OLD_BLOCK2D = 1024,     /* the identity ConvBlockRes of ResUNet level 1 (C = 32, 16 x 16 tiles) on k_resblock (one tile per block, LDS-DMA
                                      patch, swizzled rows) instead of the persistent k_block2d32 (round 6) */
  VFX_TUNE_TWO_LAUNCH_UPSAMPLERS = 2048, /* the mel ResUNet's 2 x upsamplers (odd output width) as two launches of two column phases each (one
                                      per output row class: the round-4 form) instead of one launch of four phases (round 6) */
  VFX_TUNE_DEBUG_POISON_ARENA = 256 /* debug aid, no kernel selection: the handle's workspace arena is filled with NaN patterns when
                                      it grows and before every call, so that a kernel reading a buffer nobody wrote shows up */
};

typedef struct vfx_config {
  /* front-end: config/vctk_base_voicefixer_unet.json:68-78 */
  int sample_rate;  /* 44100 */
  int n_fft;        /* 2048 (only value supported by the FFT kernels) */
  int hop;          /* 441 */
  int n_mels;       /* 128 */
  /* TFGAN vocoder layer table (third-party `voicefixer` package; see oracle/vocoder.py) */
  int voc_cond_channels;              /* 512 */
  int voc_cond_layers;                /* 5 */
  int voc_channels;                   /* 1024 */
  int voc_n_stages;                   /* 4 */
  int voc_scales[VFX_MAX_STAGES];     /* 7,7,3,3 */
  int voc_depth[VFX_MAX_STAGES];      /* 8,8,8,8 */
  int voc_dilation_base;              /* 3 */
  float voc_min_db;                   /* -115 */
  float voc_amp_floor;                /* 1e-5 */
  float voc_norm_range;               /* 4 */
  float voc_up_slope;                 /* 0.2 */
  float voc_res_slope;                /* 0.01 */
  /* arithmetic of the GEMM-shaped layers:
   * 1 (default) = split-bf16: every operand is hi + lo (two bf16), products hi*hi + hi*lo +
   *     lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (~2^-16 relative operand
   *     error; log-mel L1 vs the fp64 oracle 4e-5..8e-5, bar 1e-3);
   * 0 = exact fp32 (v_mfma_f32_32x32x2_f32), log-mel L1 5e-6..1e-5, ~1.8x slower;
   * 2 = "16-bit vocoder": the ResUNets as 1 (they carry the log-mel bar), the TFGAN vocoder on
   *     fp16 operands (the hi halves of the same layouts hold fp16 values and are the only ones
   *     loaded and multiplied: one v_mfma_f32_32x32x16_f16 per product, fp32 accumulation,
   *     saturating conversion).  BASELINE.json's 16-bit operand mode for config 2; fp16 rather
   *     than bf16 because bf16 operands hold the waveform to 40 dB SI-SDR only. */
  int precision;
  int tuning;  /* mask of VFX_TUNE_*; 0 = default */
} vfx_config;

/* Fill *cfg with the reference's hyper-parameters. */
int vfx_default_config(vfx_config* cfg);

/* Create / destroy a handle bound to HIP device `device`. */
int vfx_create(int device, const vfx_config* cfg, vfx_handle** out);
int vfx_destroy(vfx_handle* h);
const char* vfx_last_error(void);

/*
 * Weights.  `name` is the reference state_dict key (models/components/unet.py:22-53,
 * modules.py:223-261) for the UNet models, or the vocoder key convention documented in
 * oracle/vocoder.py.  `data` is a HOST pointer to contiguous fp32 of the given shape
 * (PyTorch layout).  Replaces `load_from_checkpoint` + `model.to(device)`
 * (eval_gsr_voicefixer.py:33-40).  vfx_finalize_weights folds eval-mode BatchNorm into
 * per-channel affine pairs, re-packs conv weights into the kernels' K-chunked layout and
 * uploads them.
 */
int vfx_load_tensor(vfx_handle* h, int model, const char* name, const float* data,
                    const int64_t* shape, int ndim);
int vfx_finalize_weights(vfx_handle* h, int model);

/* Workspace (bytes) a call of `model` needs for batch B and T frames; vfx_reserve grows the
 * arena up-front so that later calls never allocate (required before graph capture). */
size_t vfx_workspace_bytes(vfx_handle* h, int model, int B, int T);
int vfx_reserve(vfx_handle* h, int model, int B, int T);
/* A call made while its stream is being captured into a hipGraph pins its plan: the graph's kernel nodes hold the plan's device
 * parameter blocks and absolute pointers into the workspace arena, so from then on the handle refuses to grow (= move) its arena
 * -- a larger call returns an error instead of freeing memory the graph replays into.  vfx_unpin_plans declares that every graph
 * captured from this handle has been destroyed: plans become evictable again and the arena may grow. */
int vfx_unpin_plans(vfx_handle* h);

/*
 * STFT front-end: FDomainHelper.wav_to_spectrogram_phase (tools/pytorch/modules/
 * fDomainHelper.py:60-89) fused with MelScale.forward (tools/pytorch/mel_scale.py:52-64) and
 * optionally to_log (tools/pytorch/pytorch_util.py:157-159).
 *   wav (B, L) -> any non-NULL of: sp, cosp, sinp (B, T, 1025) ; mel (B, T, 128).
 * log10_mel != 0 writes log10(max(mel, 1e-8)) instead of linear mel.
 */
int vfx_stft_mel(vfx_handle* h, const float* wav, int B, int L, float* mel, float* sp,
                 float* cosp, float* sinp, int log10_mel, void* stream);

/* FDomainHelper.spectrogram_phase(input, eps) (tools/pytorch/modules/fDomainHelper.py:60-65) with the caller's eps:
 * mag = sqrt(clamp(re^2 + im^2, eps, inf)), cos = re / mag, sin = im / mag; any of sp / cosp / sinp (B, T, 1025) may be
 * NULL.  eps = 0 (the reference default of this method) gives 0/0 = NaN phases at exactly silent bins, as the
 * reference does; vfx_stft_mel is the eps = 1e-8 case every handler call site uses (fDomainHelper.py:67). */
int vfx_stft_phase(vfx_handle* h, const float* wav, int B, int L, float* sp, float* cosp, float* sinp,
                   float eps, void* stream);

/* MelScale.forward alone (tools/pytorch/mel_scale.py:52-64): sp (rows, 1025) -> mel (rows, 128). */
int vfx_mel_project(vfx_handle* h, const float* sp, int64_t rows, float* mel, void* stream);

/* torchlibrosa ISTFT via FDomainHelper.istft (fDomainHelper.py:30-32, used unet_v2.py:141):
 * re, im (B, T, 1025) -> wav (B, L). */
int vfx_istft(vfx_handle* h, const float* re, const float* im, int B, int T, int L, float* wav,
              void* stream);

/* Generator.forward of models/gsr_voicefixer.py:86-91 with the mel ResUNet
 * (models/components/unet.py:60-103): linear mel (B, T, 128) >= 0 -> log10 mel (B, T, 128). */
int vfx_resunet_mel(vfx_handle* h, const float* mel_linear, int B, int T, float* logmel_out,
                    void* stream);

/* UNetResComplex_100Mb.forward of models/components/unet_v2.py:86-148 (ssr_unet / gsr_unet):
 * sp (B, T, 1025), wav (B, L) -> wav_out (B, L). */
int vfx_resunet_spec(vfx_handle* h, const float* sp, const float* wav, int B, int T, int L,
                     float* wav_out, void* stream);

/* voicefixer.Vocoder.__call__ (call site eval_gsr_voicefixer.py:66):
 * linear mel (B, T, 128) -> wav (B, vfx_vocoder_out_len(T)). */
int64_t vfx_vocoder_out_len(vfx_handle* h, int T);
int vfx_vocoder(vfx_handle* h, const float* mel_linear, int B, int T, float* wav_out, void* stream);

/*
 * Whole per-segment body of handler() (eval_gsr_voicefixer.py:47-74) for a batch of
 * equal-length clips: pre -> model -> from_log -> [amp_to_original_f] -> vocoder ->
 * per-clip peak normalise -> trim_center.  wav (B, L) -> wav_out (B, L).
 * flags: bit0 = unify_energy (meta["unify_energy"], tools/utils.py:50-55).
 * logmel_out (B, T, 128) optional (may be NULL): the model's log-mel estimate.
 */
int vfx_restore_gsr(vfx_handle* h, const float* wav, int B, int L, float* wav_out,
                    float* logmel_out, int flags, void* stream);

/*
 * vfx_restore_gsr for a batch of clips of UNEQUAL length (the reference restores one file per call, of any length:
 * evaluation_proc/eval.py:119-134, eval_gsr_voicefixer.py:47-74).  wav, wav_out (B, Lmax); lengths[b] (HOST array) = samples of
 * clip b, n_fft/2 < lengths[b] <= Lmax; logmel_out (B, Lmax / hop + 1, 128) optional.  Every clip gets what its own
 * vfx_restore_gsr(B = 1, L = lengths[b]) call computes -- frames and reflect padding at its own end (fDomainHelper.py:26-28),
 * the ResUNet's zero time padding behind its own last frame (unet.py:75-77), the vocoder stopped at its own length, its own
 * peak normalisation and trim_center -- and zeros past its end in both outputs.  The clips of one call may have ANY lengths
 * (round 6): the mel ResUNet runs once per padded frame count 64 * ceil((lengths[b] / hop + 1) / 64) among them (its clips gathered
 * into a compact batch), the vocoder ONCE over the whole batch.  Hand the batch over with Lmax = (the largest padded frame count)
 * x hop - 1 when many batches follow each other: the launch plans are cached per (B, Lmax).
 */
int vfx_restore_gsr_varlen(vfx_handle* h, const float* wav, int B, int Lmax, const int* lengths, float* wav_out,
                           float* logmel_out, int flags, void* stream);

/* The spectrogram-domain twin (the per-segment body of handler_ssr_unet, eval_ssr_unet.py:77-114: sp = |STFT(wav)|, model(sp, wav),
 * models/components/unet_v2.py:86-148) for a batch of clips of unequal length: wav, wav_out (B, Lmax), lengths[b] (HOST) samples of
 * clip b.  Per clip the frames and reflect padding of its own length, the trunk's zero padding behind its own last frame, the
 * ISTFT to its own length; zeros past its end.  Same requirement: one padded frame count per call. */
int vfx_restore_ssr_varlen(vfx_handle* h, const float* wav, int B, int Lmax, const int* lengths, float* wav_out, void* stream);

/*
 * Long-audio chunkers: the two `LambdaOverlapAdd` classes (tools/dsp/overlapadd.py:337-480 and
 * tools/dsp/overlapadd_boxcar.py:338-513) segment a signal, run the network per chunk and stitch.  The
 * network call stays with the caller (all equal-length chunks as ONE batch); these are the two data
 * movements around it.
 *
 * vfx_chunk_gather = F.unfold with zero padding (overlapadd.py:421-428, overlapadd_boxcar.py:436-452):
 *   x (B, L) -> chunks (B, n_chunks, win), chunks[b][k][i] = x[b][k*hop - lead + i], 0 outside [0, L).
 * vfx_chunk_ola = synthesis window (or `scale` when window is NULL) + F.fold (overlapadd.py:455-471):
 *   frames (B, n_chunks, win) -> y (B, L), y[b][n] = sum_k frames[b][k][n + lead - k*hop] * window[..].
 */
int vfx_chunk_gather(vfx_handle* h, const float* x, int B, int L, int win, int hop, int lead,
                     int n_chunks, float* chunks, void* stream);
int vfx_chunk_ola(vfx_handle* h, const float* frames, const float* window, float scale, int B,
                  int n_chunks, int win, int hop, int lead, int L, float* y, void* stream);

/*
 * Spectral metrics of the evaluation handlers, per clip, without leaving the device
 * (evaluation_proc/metrics.py:83-95 `lsd`, `sispec`; evaluation_proc/utils.py:81-101 `energy_unify`,
 * `pow_p_norm`; used at eval_gsr_voicefixer.py:56-64):  est, target (B, T, F) -> out (B, 2),
 *   out[b][0] = LSD(est, target)    (linear-scale inputs, "mel-lsd"),
 *   out[b][1] = SiSpec(est, target) in dB (log-scale inputs for "mel-sispec", linear for "mel-non-log-sispec").
 * The reference returns the batch mean of these.
 */
int vfx_spectral_metrics(vfx_handle* h, const float* est, const float* target, int B, int T, int F,
                         float* out, void* stream);

/* Read-and-clear the sticky device flags (synchronises `stream`). */
int vfx_take_flags(vfx_handle* h, void* stream, int* flags_out);
/* ... only the bits in `mask` (VFX_FLAG_*): the others stay raised for a later check. */
int vfx_take_flags_masked(vfx_handle* h, void* stream, int mask, int* flags_out);

/*
 * Turns for work the library does not enqueue itself.  Calls of this library on different streams of one device take turns on
 * the GPU timeline (see "Conventions"); a call made during a stream capture is exempt, so the REPLAY of a hipGraph captured
 * from such calls must be bracketed by the caller when anything of this library may run on another stream of the device at
 * the same time:  vfx_turn_begin(device, s); hipGraphLaunch(graph, s); vfx_turn_end(device, s);  (Engine.replay does this).
 * begin makes `s` wait for the end of the device's previous turn, end leaves the event the next call waits for.  Not needed
 * when everything of a process runs on one stream per device.
 */
int vfx_turn_begin(int device, void* stream);
int vfx_turn_end(int device, void* stream);

/*
 * Live kernel timing for the roofline report: between vfx_profile_begin and vfx_profile_end
 * every tap-convolution launch is bracketed by HIP events on its own stream.
 * vfx_profile_end synchronises and returns the number of launches, the sum of their
 * durations (ms) and of their algorithmic FLOPs (2*M*Cout*K); out pointers may be NULL.
 */
int vfx_profile_begin(vfx_handle* h);
int vfx_profile_end(vfx_handle* h, int64_t* launches, double* total_ms, double* total_flops);

#ifdef __cplusplus
}
#endif
#endif /* VFX_H_ */
