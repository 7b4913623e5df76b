#!/usr/bin/env python3
"""bench.py -- VoiceFixer 44.1 kHz restoration throughput on MI355X.

Metric (BASELINE.json): restored-audio seconds per wall-second (RTF^-1).

    python bench.py [--gpus N --steps K --warmup W] [--workload NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with no WORLD_SIZE in the environment re-executes itself under torch.distributed.run with N ranks
(one process per GPU, 127.0.0.1 rendezvous); a WORLD_SIZE that disagrees with --gpus is an error.

Workloads (BASELINE.json `configs`; a "step" is one pass of the hot path over one batch of synthetic clips that are
resident in HBM when the clock starts):

  gsr16x10     (default, configs[1]) gsr_voicefixer restore -- STFT -> mel -> ResUNet -> from_log -> TFGAN vocoder ->
               peak normalise -> trim (eval_gsr_voicefixer.py:47-74) -- of 16 x 10 s clips per GPU.  Weak scaling:
               every rank restores its own shard, no collective on the data path.
  sharded1024  (configs[3]) the same restore over 128 clips per GPU (1024 on 8 GPUs) that live on rank 0: scatter over
               RCCL point-to-point -> per-rank restore (sub-batched) -> gather, ALL inside the timed region.
  ssr_sr64     (configs[2]) ssr_unet super-resolution forward, 64 x 3 s per GPU: |STFT| -> spectrogram ResUNet (second
               STFT for the phase, trunk, recombination, ISTFT) (eval_ssr_unet.py:77-114, unet_v2.py:86-148).
  stream1s     (configs[4]) gsr_unet streaming: one 1-s chunk per step, the step captured in a hipGraph.

Arithmetic (--precision): 2 (default) = BASELINE.json's 16-bit operand mode for configs[1]: the
ResUNet on split-bf16 operands (it carries the log-mel L1 <= 1e-3 bar), the vocoder on fp16 operands with one MFMA per
product; 1 = split-bf16 everywhere (3 MFMAs per product), timed as well at N = 1 and reported beside `value` as
`split_bf16_mode`; 0 = exact fp32 MFMA.

Rank 0 prints ONE JSON line.  At N = 1 the default run also times configs[2] and configs[4] for a few steps in the same
process and nests their value / ms_per_step / parity (vs the float64 oracle) / roofline under `aux_workloads`, together
with `varlen_vctk`: 128 clips of 2 .. 8 s, no two of one length, through the padded-batch entry point (--no-aux skips them).  Measured in the same run: `parity` (HIP outputs of the benched batch vs the CPU oracle on
the clips the oracle was run on; the run FAILS when the log-mel L1 exceeds the 1e-3 bar), `roofline` (HIP events around
every convolution launch on its own stream over K more steps; `traffic` from two rocprofv3 PMC passes of a child run of
the same workload), `roofline_hbm` (the HBM-bound front-end / back-end kernels), `cpu_baseline` (the CPU oracle, a port
of the reference algorithm, on a bounded sample of the same clips on this box's host cores).
"""
import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA (spec; 2495 measured)
PEAK_VALU_TFLOPS = 157.3        # fp32 vector peak (packed FMA), MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MAC_MEL_PER_FRAME = 92853664
MAC_SPEC_PER_FRAME = 780251136
LOGMEL_L1_BAR = 1e-3            # BASELINE.json north_star: mel L1 <= 1e-3 vs the reference forward
WORKLOADS = ("gsr16x10", "sharded1024", "ssr_sr64", "stream1s")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=WORKLOADS, default="gsr16x10")
    ap.add_argument("--clips", type=int, default=0, help="clips per GPU per step (default: the workload's)")
    ap.add_argument("--seconds", type=float, default=0.0, help="clip length (default: the workload's)")
    ap.add_argument("--cpu-baseline-clips", type=int, default=-1, help="clips in the CPU-oracle sample (0 = skip; default per workload)")
    ap.add_argument("--cpu-threads", type=int, default=32)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle comparison (needs the cpu baseline sample)")
    ap.add_argument("--traffic", choices=("live", "off"), default="live",
                    help="live: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of a child run of the same workload")
    ap.add_argument("--dist-selfcheck", action="store_true", help="N > 1: round-trip a tensor through dist.scatter_clips / gather_clips first")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra all-split-bf16 (precision 1) timing at N = 1")
    ap.add_argument("--no-aux", action="store_true", help="skip the nested ssr_sr64 / stream1s measurements of the default run")
    ap.add_argument("--aux-steps", type=int, default=3)
    ap.add_argument("--cpu-repeats", type=int, default=3, help="timed calls of the CPU oracle (the median is reported)")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="TEST ONLY (tests/test_host.py): run this script's own rank plumbing -- process group, weight broadcast, "
                         "sharding, barrier + max-over-ranks clock, the JSON line -- on CPU over gloo with the engine replaced by "
                         "an identity stub; the line says dry_run_cpu and its value means nothing")
    ap.add_argument("--tuning", type=int, default=0, help="vfx_config.tuning mask (include/vfx.h VFX_TUNE_*); 0 = the shipped kernel selection")
    ap.add_argument("--precision", type=int, default=2, help="0 = exact fp32 MFMA, 1 = split-bf16 (hi+lo, 3 bf16 MFMAs), 2 = ResUNet split-bf16 + vocoder fp16 (1 MFMA per product)")
    args = ap.parse_args()
    d_clips, d_sec, d_cpu = {"gsr16x10": (16, 10.0, 2), "sharded1024": (128, 10.0, 0), "ssr_sr64": (64, 3.0, 1),
                             "stream1s": (1, 1.0, 1)}[args.workload]
    args.clips = args.clips or d_clips
    args.seconds = args.seconds or d_sec
    if args.cpu_baseline_clips < 0:
        args.cpu_baseline_clips = d_cpu
    return args


def maybe_spawn(args):
    """--gpus N without a launcher: become `torch.distributed.run` with N ranks of this very command."""
    ws = os.environ.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != args.gpus:
            sys.exit("bench.py: --gpus %d but WORLD_SIZE=%s -- launch N ranks with --gpus N (or drop WORLD_SIZE and let "
                     "bench.py spawn them)" % (args.gpus, ws))
        return
    if args.gpus <= 1:
        return
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


# ---------------------------------------------------------------------------------------------------------
# parity (HIP vs the CPU oracle, measured in the run)
# ---------------------------------------------------------------------------------------------------------
def sisdr_db(est, ref):
    est, ref = np.asarray(est, np.float64), np.asarray(ref, np.float64)
    err = est - ref
    return float(10.0 * np.log10((ref ** 2).sum() / ((err ** 2).sum() + 1e-30)))


def parity_gsr(out, logmel, ref):
    """out (n, L), logmel (n, T, 128) of the HIP path vs the oracle's dict for the same n clips."""
    lm, rl = logmel.cpu().numpy().astype(np.float64), ref["logmel"][:, 0].astype(np.float64)
    w, rw = out.cpu().numpy(), ref["wav"][:, 0]
    return {"logmel_l1": float(np.abs(lm - rl).mean()), "logmel_max": float(np.abs(lm - rl).max()),
            "wav_sisdr_db": round(sisdr_db(w, rw), 2), "wav_max_err": float(np.abs(w - rw).max()), "clips": int(w.shape[0]),
            "frames": int(lm.shape[1]), "vs": "oracle.pipeline.restore_gsr (fp32 CPU port of the reference forward)",
            "bar": {"logmel_l1": LOGMEL_L1_BAR}}


def parity_wav(out, ref_wav):
    w = out.cpu().numpy()
    return {"wav_sisdr_db": round(sisdr_db(w, ref_wav), 2), "wav_max_err": float(np.abs(w - ref_wav).max()),
            "clips": int(w.shape[0]), "vs": "oracle.pipeline.restore_ssr (fp32 CPU port of unet_v2.py:86-148)"}


# ---------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle timed on this box's host cores; its outputs double as the parity reference)
# ---------------------------------------------------------------------------------------------------------
def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def sample_indices(B, n):
    """The clips of a batch of B that the CPU oracle runs on: the first ceil(n/2) and the LAST floor(n/2) -- so that the
    parity of the benched batch covers both ends of it (the clips in between are covered HIP-vs-HIP, `batch_tail_equal`)."""
    n = max(0, min(n, B))
    head = (n + 1) // 2
    return list(range(head)) + list(range(B - (n - head), B))


def cpu_baseline(kind, clips, n_clips, threads, repeats=3, dtype=None):
    """the oracle (a port of the reference algorithm) on the box's host cores, after one warm-up
    call, MEDIAN of `repeats` timed calls on a bounded sample of the benched clips; thread count and CPU model stated.
    The oracle's torch-CPU convolutions stop scaling (and then collapse) beyond a few dozen threads at these sizes
    (2 x EPYC 9575F host: 1-s clip 0.23 / 0.18 / 0.25 / 0.66 s at 8 / 16 / 32 / 64 threads, minutes at 256), so the
    baseline uses a fixed, stated thread count instead of every hardware thread.  `dtype` = torch.float64: the ssr oracle
    as the PARITY reference (two fp32 evaluations of that trunk agree to ~60 dB only); it is then timed once."""
    from oracle import pipeline
    from voicefixer_main_amd import synth
    wav = clips[sample_indices(clips.shape[0], n_clips)]
    threads = max(1, min(threads, os.cpu_count() or 1))
    torch.set_num_threads(threads)
    times = []
    if kind == "gsr":
        unet_sd, voc_sd = synth.make_resunet_state_dict(0), synth.make_vocoder_state_dict(1)
        pipeline.restore_gsr(unet_sd, voc_sd, wav[:1, :, :44100])          # warm-up (thread pool, allocator)
        for _ in range(max(1, repeats)):
            t0 = time.perf_counter()
            ref = pipeline.restore_gsr(unet_sd, voc_sd, wav)
            times.append(time.perf_counter() - t0)
        name = "oracle.pipeline.restore_gsr"
    else:
        unet_sd = synth.make_resunet_state_dict(2)
        pipeline.restore_ssr(unet_sd, wav[:1, :, :22050])
        for _ in range(max(1, repeats)):
            t0 = time.perf_counter()
            ref = pipeline.restore_ssr(unet_sd, wav)
            times.append(time.perf_counter() - t0)
        name = "oracle.pipeline.restore_ssr"
        if dtype is not None:    # the parity reference in float64 (not the timed baseline)
            sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in unet_sd.items()}
            ref = pipeline.restore_ssr(sd64, wav, dtype=dtype)
    dt = float(np.median(times))
    seconds = wav.shape[0] * wav.shape[-1] / 44100.0
    return {"value": round(seconds / dt, 3), "unit": "audio-s/s", "cores": threads, "kind": "port",
            "seconds": round(dt, 2), "timed_calls": [round(t, 2) for t in times], "statistic": "median",
            "host_cpus": os.cpu_count(), "cpu_model": cpu_model(),
            "sample": "%s (torch-CPU fp32 + numpy port of the reference algorithm) on %d clip(s) x %.0f s of the same "
                      "synthetic clips (batch indices %s) as one batch, same seeded weights, after a warm-up call"
                      % (name, wav.shape[0], wav.shape[-1] / 44100.0, sample_indices(clips.shape[0], n_clips))}, ref


# ---------------------------------------------------------------------------------------------------------
# roofline: MFMA-bound convolution kernels (HIP events inside libvfx), HBM-bound DSP kernels, PMC traffic
# ---------------------------------------------------------------------------------------------------------
def measure_conv_roofline(eng, step, args, traffic, ms_step=None, steps=None):
    """`ms_step` = the timed region's ms per step (for the whole-step MFMA line).
    HIP events around every convolution launch (on the launch stream) over K more steps of the same workload;
    per-kernel totals come from the per-launch table libvfx writes (VFX_PROFILE_DUMP: duration, algorithmic flops and
    algorithmic HBM bytes of every launch).  The roofline object describes the kernel with the largest share of GPU time
    against the roofline that bounds it (its algorithmic intensity vs the ridge of the chip); both lines are reported."""
    import csv
    keep = os.environ.get("VFX_PROFILE_DUMP")   # a caller-provided path keeps the per-launch table
    dump = keep or tempfile.NamedTemporaryFile(prefix="vfx_convs_", suffix=".csv", delete=False).name
    os.environ["VFX_PROFILE_DUMP"] = dump
    steps = max(steps or args.steps, 1)
    eng.profile_begin()
    for _ in range(steps):
        step()
    n, ms, fl = eng.profile_end()
    rows = list(csv.DictReader(open(dump)))
    if not keep:
        os.environ.pop("VFX_PROFILE_DUMP", None)
        os.unlink(dump)
    per = {}
    for r in rows:
        k = r["kernel"].replace(";", ",")
        t = per.setdefault(k, [0, 0.0, 0.0, 0.0, 0.0])
        t[0] += 1
        t[1] += float(r["ms"])
        t[2] += float(r["tflops"]) * float(r["ms"]) * 1e9   # flops of the launch
        t[3] += float(r.get("bytes", 0) or 0)               # algorithmic HBM bytes of the launch
        t[4] += float(r.get("design_bytes", 0) or 0)        # bytes the kernel's own data layout moves
    dom = max(per, key=lambda k: per[k][1])
    cnt, kms, kfl, kby, kdb = per[dom]
    tflops = kfl / (kms * 1e-3) / 1e12
    gbs = kby / (kms * 1e-3) / 1e9
    split = args.precision >= 1
    peak = PEAK_BF16_MFMA_TFLOPS if split else PEAK_FP32_MFMA_TFLOPS
    plain = "f16" in dom.split(">")[-1]      # a 16-bit launch of the precision-2 vocoder: one MFMA per product
    per_product = 1 if (plain or not split) else 3
    tr = (traffic or {}).get("kernels", {}).get(dom)
    # Which roofline bounds the kernel: its algorithmic intensity (flops the MFMA pipe must issue per algorithmic HBM
    # byte) against the ridge of the chip, peak MFMA rate / peak HBM rate.
    intensity = kfl * per_product / max(kby, 1.0)
    ridge = peak * 1e12 / (PEAK_HBM_GBS * 1e9)
    hbm_bound = intensity < ridge
    mfma_line = {"achieved": round(tflops, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tflops / peak, 4),
                 # achieved counts ALGORITHMIC flops (2*M*N*K once); in split-bf16 mode the kernel issues 3 bf16 MFMAs per
                 # product, so `frac` is bounded by 1/3 and mfma_issue_frac is the share of the MFMA pipe actually used
                 "mfma_issue_frac": round(tflops * per_product / peak, 4)}
    hbm_line = {"achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                # a ResStack layer = x in + y out (4 bytes per element on the fp16 trunk, 8 on the fp32
                # one); a convolution = its sources, residual and outputs once.  design_bytes = what the kernel's data layout moves
                # (round 3's two-form trunk: 12 bytes per element at C = 256; on the fp16 trunk = the algorithmic bytes).
                "algorithmic_bytes_per_launch": round(kby / max(cnt, 1)), "design_bytes_per_launch": round(kdb / max(cnt, 1))}
    head = dict(hbm_line if hbm_bound else mfma_line)
    return dict({
        "bound": "hbm" if hbm_bound else "mfma",
        "accounting": "SURVEY.md section 8(d): algorithmic flops = 2 x MAC once; algorithmic bytes = tensors in + out once "
                      "(ResStack layer: x in + y out = 4 B per element on the fp16 trunk, 8 B on the fp32 trunk of "
                      "VFX_TUNE_F32_TRUNK); the bound follows from that intensity vs the chip's ridge",
        "kernel": "%s (%s)" % (dom, "1 x v_mfma_f32_32x32x16_f16 per product (fp16 operands), fp32 accumulate" if plain
                               else "3 x v_mfma_f32_32x32x16_bf16 per product (hi*hi + hi*lo + lo*hi), fp32 accumulate"
                               if split else "v_mfma_f32_32x32x2_f32"),
    }, **head, **{
        "intensity_flop_per_byte": round(intensity, 1), "ridge_flop_per_byte": round(ridge, 1),
        "mfma_roofline": mfma_line, "hbm_roofline": hbm_line,
        "traffic": tr["bytes_per_launch"] if tr else None,
        "traffic_unit": "HBM bytes per launch: 2 * FETCH_SIZE + WRITE_SIZE of this run's rocprofv3 PMC passes",
        "traffic_detail": tr,
        "traffic_error": (traffic or {}).get("error"),
        "launches_per_step": cnt // steps,
        "avg_launch_us": round(kms * 1e3 / max(cnt, 1), 2),
        "kernel_ms_per_step": round(kms / steps, 3),
        "algorithmic_gflop_per_step": round(kfl / steps / 1e9, 1),
        "share_of_conv_time": round(kms / max(ms, 1e-9), 4),
        "all_conv_kernels": {k: {"launches_per_step": v[0] // steps, "ms_per_step": round(v[1] / steps, 3),
                                 "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 1),
                                 "algorithmic_gbs": round(v[3] / (v[1] * 1e-3) / 1e9, 1),
                                 "frac_mfma": round(v[2] / (v[1] * 1e-3) / 1e12 / peak, 4),
                                 "frac_hbm": round(v[3] / (v[1] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                 "hbm_bytes_per_launch": ((traffic or {}).get("kernels", {}).get(k) or {}).get("bytes_per_launch")}
                             for k, v in sorted(per.items())},
        # the next kernels by share of GPU time, each against the MFMA peak: after round 5 the top two groups are within 4 % of each
        # other (k_conv<128> f16 = 25 heterogeneous launches -- the C = 512 stack, the upsamplers, the condnet; k_resblock<256, 4> f16
        # = the eight C = 256 layers), so which one "dominates" can change from box to box
        "runners_up": [{"kernel": k, "ms_per_step": round(v[1] / steps, 3), "avg_launch_us": round(v[1] * 1e3 / max(v[0], 1), 2),
                        "frac_mfma": round(v[2] / (v[1] * 1e-3) / 1e12 / peak, 4),
                        "hbm_bytes_per_launch": ((traffic or {}).get("kernels", {}).get(k) or {}).get("bytes_per_launch")}
                       for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[1:3]],
        "all_conv_ms_per_step": round(ms / steps, 3),
        "all_conv_algorithmic_gflop_per_step": round(fl / steps / 1e9, 1),
    }), step_line(fl / steps, ms_step, peak)


def step_line(flop_per_step, ms_step, peak):
    """The whole step against the dense 16-bit MFMA peak: algorithmic flops of every GEMM-shaped launch / the timed
    region's time per step."""
    if not ms_step:
        return None
    tf = flop_per_step / (ms_step * 1e-3) / 1e12
    return {"gflop": round(flop_per_step / 1e9, 1), "ms": round(ms_step, 3), "tflops": round(tf, 1), "peak": peak,
            "frac_of_mfma_peak": round(tf / peak, 4)}


def measure_hbm_stages(eng, B, L, reps=20):
    """The HBM-bound kernels of the path against the 8 TB/s roofline, algorithmic bytes per frame from SURVEY.md
    section 8(d): mel-only front-end 441*4 + 128*4 = 2276 B, phase-emitting front-end 441*4 + 3*1025*4 = 14064 B,
    ISTFT 2*1025*4 + 441*4 = 9964 B.  Timed with events on the stream the kernels run on (torch's current stream is
    the one handed to libvfx)."""
    dev = eng.device
    wav = torch.randn((B, L), device=dev) * 0.1
    T = L // 441 + 1
    frames = B * T

    def timed(fn):
        fn()
        torch.cuda.synchronize(dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize(dev)
        return a.elapsed_time(b) / reps * 1e-3

    mel = torch.empty((B, T, 128), device=dev)
    sp, co, si = (torch.empty((B, T, 1025), device=dev) for _ in range(3))
    out = torch.empty((B, L), device=dev)
    import ctypes
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    lib, h, st = eng.lib, eng.h, eng._stream()
    t_mel = timed(lambda: lib.vfx_stft_mel(h, P(wav), B, L, P(mel), None, None, None, 0, st))
    t_ph = timed(lambda: lib.vfx_stft_mel(h, P(wav), B, L, None, P(sp), P(co), P(si), 0, st))
    re, im = sp * co, sp * si
    t_is = timed(lambda: lib.vfx_istft(h, P(re), P(im), B, T, L, P(out), st))

    # arithmetic per frame (fp32 flops): 1024-point complex FFT 5 N log2 N = 51 200; real-FFT untangle + magnitude 1025 x 21
    # (+ 3 for cos / sin); window 2048; mel 2 x 2018.  Inverse: pack 1024 x 20, FFT, window + scale 2 x 2048, overlap-add 2048.
    # Against the fp32 vector peak (157.3 TFLOP/s, packed FMA) the ridge is 19.7 flop/B: the mel-only front-end (35 flop/B)
    # is bound by VALU issue, the phase-emitting one and the ISTFT by HBM.
    def line(bytes_per_frame, flop_per_frame, t):
        gbs = bytes_per_frame * frames / t / 1e9
        tf = flop_per_frame * frames / t / 1e12
        inten = flop_per_frame / bytes_per_frame
        return {"bound": "valu" if inten > PEAK_VALU_TFLOPS * 1e3 / PEAK_HBM_GBS else "hbm",
                "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                "valu": {"achieved": round(tf, 2), "peak": PEAK_VALU_TFLOPS, "unit": "TFLOP/s (fp32 vector)",
                         "frac": round(tf / PEAK_VALU_TFLOPS, 4), "flop_per_frame": flop_per_frame},
                "intensity_flop_per_byte": round(inten, 1),
                "bytes_per_frame": bytes_per_frame, "frames": frames, "us": round(t * 1e6, 1)}
    fft = 51200
    return {"stft_mel": line(2276, fft + 1025 * 21 + 2048 + 2 * 2018, t_mel),
            "stft_phase": line(14064, fft + 1025 * 24 + 2048, t_ph),
            "istft": line(9964, fft + 1024 * 20 + 2 * 2048 + 2048, t_is)}


def live_traffic(args):
    """HBM bytes per launch of every convolution kernel from two rocprofv3 PMC passes (FETCH_SIZE; WRITE_SIZE -- they
    do not fit one pass) of a CHILD run of this workload (1 warm-up + 1 step, kernel-trace only).  bytes = 2 *
    FETCH_SIZE + WRITE_SIZE (KiB): on gfx950 FETCH_SIZE reports half of the bytes of a wide streaming read
    (MI355X_MICROARCH.md, HBM section).  Returns {"kernels": {...}} or {"error": ...}."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    sys.path.insert(0, os.path.join(HERE, "scripts"))
    import collections
    import sqlite3
    from kname import short
    out = tempfile.mkdtemp(prefix="vfx_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "VFX_PROFILE_DUMP"):
        env.pop(k, None)
    child = [sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--no-roofline", "--no-alt",
             "--cpu-baseline-clips", "0", "--traffic", "off", "--no-parity", "--no-aux", "--workload", args.workload,
             "--precision", str(args.precision), "--clips", str(args.clips), "--seconds", str(args.seconds)]
    try:
        dbs = {}
        for name, counter in (("tcc1", "FETCH_SIZE"), ("tcc2", "WRITE_SIZE")):
            # GRBM_GUI_ACTIVE rides along with the read pass: cycles the GPU was busy during a dispatch, summed over the 8 XCDs
            ctrs = [counter, "GRBM_GUI_ACTIVE"] if name == "tcc1" else [counter]
            r = subprocess.run([exe, "--kernel-trace", "--pmc"] + ctrs + ["-d", os.path.join(out, name), "-o", name, "--"] + child,
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            found = [os.path.join(d, f) for d, _, fs in os.walk(os.path.join(out, name)) for f in fs if f.endswith("_results.db")]
            if r.returncode != 0 or not found:
                return {"error": "rocprofv3 --pmc %s failed (rc %d): %s" % (counter, r.returncode, (r.stderr or r.stdout)[-300:])}
            dbs[counter] = found[0]

        def load(db, counter):
            c = sqlite3.connect(db)
            rows = c.execute("select dispatch_id, kernel_name, sum(value) from counters_collection where counter_name=? "
                             "group by dispatch_id order by dispatch_id", (counter,)).fetchall()
            return [(short(k), v) for _, k, v in rows]
        rd, wr = load(dbs["FETCH_SIZE"], "FETCH_SIZE"), load(dbs["WRITE_SIZE"], "WRITE_SIZE")
        # effective shader clock per dispatch = busy cycles (per XCD) / wall time of the dispatch (MI355X_MICROARCH.md, DVFS)
        clk = {}
        try:
            c = sqlite3.connect(dbs["FETCH_SIZE"])
            for did, k, v, st, en in c.execute("select dispatch_id, kernel_name, sum(value), min(start), max(end) from "
                                               "counters_collection where counter_name='GRBM_GUI_ACTIVE' group by dispatch_id"):
                if en > st:
                    clk.setdefault(short(k), []).append(v / 8.0 / (en - st))
        except Exception:
            clk = {}
        if len(rd) != len(wr) or not rd:
            return {"error": "PMC passes disagree on the launch sequence (%d vs %d)" % (len(rd), len(wr))}
        first = [i for i, (k, _) in enumerate(rd) if k.startswith("k_stft_mel")]
        last = first[-1] if first else 0                      # launches of the last step
        per = collections.OrderedDict()
        for (k, r), (_, w) in list(zip(rd, wr))[last:]:
            if not (k.startswith("k_conv") or k.startswith("k_resblock") or k.startswith("k_up16")):
                continue
            t = per.setdefault(k, [0, 0.0, 0.0])
            t[0] += 1
            t[1] += 2.0 * r * 1024.0
            t[2] += w * 1024.0
        def ghz(k):
            g = sorted(clk.get(k, []))
            g = g[len(g) // 2] if g else None
            return round(g, 3) if g and 0.3 < g < 2.7 else None
        return {"kernels": {k: {"launches_per_step": v[0], "read_bytes_per_launch": round(v[1] / v[0]),
                                "write_bytes_per_launch": round(v[2] / v[0]), "bytes_per_launch": round((v[1] + v[2]) / v[0]),
                                "effective_sclk_ghz_profiled": ghz(k)}
                            for k, v in per.items()}}
    except Exception as e:  # never fatal for the throughput measurement
        return {"error": repr(e)[:300]}
    finally:
        shutil.rmtree(out, ignore_errors=True)


class PowerSampler:
    """Package power and shader clock of the GPU while the timed region runs, read from the amdgpu hwmon files (a thread that
    polls sysfs every few ms: no rocm-smi process, nothing on the GPU).  The numbers belong next to every fraction of a peak
    that assumes 2.4 GHz.  sysfs lists every GPU of the node while the container sees one: the card is
    matched by PCI address, or -- without one -- taken as the card that drew the most power over the region.  Absent files (no
    GPU, other driver) give None."""

    def __init__(self, index=0):
        import glob
        self.cards, self.samples, self._stop, self._thr, self.match = [], {}, False, None, None
        for c in sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*")):
            f = {}
            for key, names in (("power_uw", ("power1_average", "power1_input")), ("sclk_hz", ("freq1_input",)), ("cap_uw", ("power1_cap",))):
                for n in names:
                    if os.path.exists(os.path.join(c, n)):
                        f[key] = os.path.join(c, n)
                        break
            if "sclk_hz" in f or "power_uw" in f:
                self.cards.append((c, f))
                self.samples[c] = []
        try:
            pr = torch.cuda.get_device_properties(index)
            addr = "%04x:%02x:%02x." % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            self.match = next((c for c, _ in self.cards if addr in os.path.realpath(os.path.join(c, "..", ".."))), None)
        except Exception:
            self.match = None

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except (OSError, ValueError, TypeError):
            return None

    def __enter__(self):
        if self.cards:
            import threading
            watch = [(c, f) for c, f in self.cards if self.match in (None, c)]

            def run():
                while not self._stop:
                    for c, f in watch:
                        self.samples[c].append((self._read(f.get("power_uw")), self._read(f.get("sclk_hz"))))
                    time.sleep(0.004)
            self._thr = threading.Thread(target=run, daemon=True)
            self._thr.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._thr:
            self._thr.join(timeout=1.0)

    def result(self):
        best = None
        for c, f in self.cards:
            smp = self.samples.get(c) or []
            pw = [p for p, _ in smp if p]
            ck = [k for _, k in smp if k]
            if not pw and not ck:
                continue
            avg = sum(pw) / len(pw) if pw else 0.0
            if best is None or avg > best[0]:
                best = (avg, c, f, pw, ck, len(smp))
        if best is None:
            return None
        _, c, f, pw, ck, n = best
        cap = self._read(f.get("cap_uw"))
        return {"samples": n, "avg_power_w": round(sum(pw) / len(pw) / 1e6, 1) if pw else None,
                "max_power_w": round(max(pw) / 1e6, 1) if pw else None, "power_cap_w": round(cap / 1e6, 1) if cap else None,
                "avg_sclk_mhz": round(sum(ck) / len(ck) / 1e6, 1) if ck else None,
                "min_sclk_mhz": round(min(ck) / 1e6, 1) if ck else None, "nominal_sclk_mhz": 2400,
                "card": os.path.basename(os.path.dirname(os.path.dirname(os.path.dirname(c)))),
                "card_matched_by": "pci address" if self.match else "highest average power over the region",
                "source": "amdgpu hwmon (power1_average / freq1_input), polled every 4 ms over the timed region"}


# ---------------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------------
class _StubEngine:
    """--dry-run-cpu: stands where the libvfx handle stands, computes nothing (identity).  Never used by a measurement."""

    def __init__(self, device, config=None):
        self.device, self.loaded = torch.device(device), {}

    def load_state_dict(self, model, sd, prefix=""):
        self.loaded[model] = sum(int(v.numel()) for v in sd.values())

    def restore_gsr(self, wav, unify_energy=False, want_logmel=False, out=None):
        res = wav.clone() if out is None else out.copy_(wav)
        return (res, torch.zeros((wav.shape[0], wav.shape[1] // 441 + 1, 128))) if want_logmel else res

    def take_flags(self, mask=None):
        return 0


class Workload:
    """One BASELINE.json config on one rank: `step()` enqueues one pass of the hot path over its batch of clips, which are
    resident in HBM before the clock starts."""

    def __init__(self, wl, args, device, rank, world, precision, clips_n=None, seconds=None, weights=None, tuning=0):
        from voicefixer_main_amd import dist as vdist
        from voicefixer_main_amd import synth
        from voicefixer_main_amd.engine import Engine, MODEL_UNET_MEL, MODEL_UNET_SPEC, MODEL_VOCODER
        self.cuda = torch.device(device).type == "cuda"
        if not self.cuda:          # --dry-run-cpu
            assert args.dry_run_cpu and wl in ("gsr16x10", "sharded1024")
            Engine = _StubEngine
        d_clips, d_sec = {"gsr16x10": (16, 10.0), "sharded1024": (128, 10.0), "ssr_sr64": (64, 3.0), "stream1s": (1, 1.0)}[wl]
        self.wl, self.device, self.rank, self.world, self.precision = wl, device, rank, world, precision
        self.B = B = clips_n or d_clips
        self.seconds = seconds = seconds or d_sec
        self.gsr = wl in ("gsr16x10", "sharded1024")
        self.extra = {}
        self.phases = None
        self.eager = None
        self.weights = weights if weights is not None else {}
        self.tuning = tuning
        self.step_ms = []

        def state_dict(seed, kind):
            # N ranks: rank 0 builds (or, in production, reads the checkpoint) once, everybody else receives ONE flat
            # broadcast -- dist.broadcast_state_dict; a world of one builds locally.
            key = (seed, kind)
            if key not in self.weights:
                sd = None
                if rank == 0:
                    sd = synth.make_resunet_state_dict(seed) if kind == "unet" else synth.make_vocoder_state_dict(seed)
                self.weights[key] = vdist.broadcast_state_dict(sd, device)
            return self.weights[key]

        def make_engine(precision, tuning=tuning):
            e = Engine(device, config={"precision": precision, "tuning": tuning})
            if self.gsr:
                e.load_state_dict(MODEL_UNET_MEL, state_dict(0, "unet"))
                e.load_state_dict(MODEL_VOCODER, state_dict(1, "voc"))
            else:
                e.load_state_dict(MODEL_UNET_SPEC, state_dict(2, "unet"))
            return e
        self.make_engine = make_engine
        self.eng = eng = make_engine(precision)

        if wl == "gsr16x10":
            self.clips = synth.make_clips(B, seconds, seed=1234 + 1000 * rank)               # (B, 1, L) float32, host
            self.wav = wav = torch.from_numpy(self.clips[:, 0]).to(device)                   # resident in HBM
            self.out = out = torch.empty_like(wav)
            self.L = wav.shape[1]
            self.step = lambda e=eng: e.restore_gsr(wav, out=out)
            self.audio_per_step = world * B * seconds
        elif wl == "sharded1024":
            self.n_total = n_total = B * world
            self.L = L = int(round(seconds * 44100))
            if rank == 0:
                self.clips = synth.make_clips(n_total, seconds, seed=1234)
                self.full = full = torch.from_numpy(self.clips[:, 0]).to(device)             # all clips live on rank 0
            else:
                self.clips, self.full = None, None
                full = None
            self.phases = phases = {"scatter_ms": 0.0, "restore_ms": 0.0, "gather_ms": 0.0}
            self.gathered = gathered = [None]
            sync = (lambda: torch.cuda.synchronize(device)) if self.cuda else (lambda: None)

            def step(e=eng, acc=phases):
                back, t = vdist.sharded_step(lambda x: e.restore_gsr(x), full, n_total, L, device, sync=sync)
                gathered[0] = back
                for k in acc:
                    acc[k] += t[k]
            self.step = step
            self.audio_per_step = n_total * seconds
        elif wl == "ssr_sr64":
            # 1-kHz cheby1 low-pass: 2 kHz -> 44.1 kHz super-resolution input
            self.clips = synth.make_clips(B, seconds, seed=7 + 1000 * rank, mode="lowpass")
            self.wav = wav = torch.from_numpy(self.clips[:, 0]).to(device)
            self.L = wav.shape[1]
            self.holder = holder = [None]

            def step(e=eng):
                sp = e.stft(wav, want_mel=False, want_sp=True)["sp"]        # eval_ssr_unet.py:80
                holder[0] = e.resunet_spec(sp, wav)                         # STFT (phase) + trunk + ISTFT, unet_v2.py:86-148
            self.step = step
            self.audio_per_step = world * B * seconds
        else:  # stream1s
            self.clips = synth.make_clips(1, seconds, seed=11 + 1000 * rank)
            self.wav = wav = torch.from_numpy(self.clips[:, 0]).to(device)
            self.L = wav.shape[1]
            self.holder = holder = [None]

            def eager(e=eng):
                sp = e.stft(wav, want_mel=False, want_sp=True)["sp"]
                holder[0] = e.resunet_spec(sp, wav)
            eager()                                                         # plans, arena (no allocation inside the capture)
            side = torch.cuda.Stream(device)
            side.wait_stream(torch.cuda.current_stream(device))
            self.graph = graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                eager()
                torch.cuda.synchronize(device)
                with torch.cuda.graph(graph, stream=side):
                    eager()
            torch.cuda.current_stream(device).wait_stream(side)
            self.step = graph.replay
            self.eager = eager
            self.audio_per_step = world * seconds
            self.extra["hipgraph"] = True

    def timed(self, steps, warmup, barrier=lambda: None):
        """W untimed steps, then EXACTLY `steps` steps between barrier + synchronize on both sides -> seconds."""
        sync = (lambda: torch.cuda.synchronize(self.device)) if self.cuda else (lambda: None)
        for _ in range(warmup):
            self.step()
        sync()
        if self.phases:
            for k in self.phases:
                self.phases[k] = 0.0
        # one event per step boundary on the launch stream (torch's current stream is the one handed to libvfx): the spread of
        # the K steps inside the SAME timed region -- the clock around the region stays the host's, as the contract says
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if self.cuda else None
        barrier()
        t0 = time.perf_counter()
        if ev:
            ev[0].record()
        for i in range(steps):
            self.step()
            if ev:
                ev[i + 1].record()
        sync()
        barrier()
        dt = time.perf_counter() - t0
        self.step_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)] if ev else []
        return dt

    def flags(self):
        """The handle's sticky flags after the timed region: a 16-bit vocoder whose activations left the fp16 range, or a
        negative mel, makes the measurement INVALID (bench.py times the raw entry point, not the re-running wrapper)."""
        from voicefixer_main_amd import _lib
        f = self.eng.take_flags()
        return {"negative_input": bool(f & _lib.FLAG_NEGATIVE_INPUT), "f16_saturated": bool(f & _lib.FLAG_F16_SATURATED)}


F32_TRUNK = 64   # include/vfx.h VFX_TUNE_F32_TRUNK


def dtype_string(gsr, precision, tuning=0):
    trunk = "fp32 residual trunk between the ResStack launches" if tuning & F32_TRUNK else \
        "fp16 residual trunk between the ResStack launches (sums in fp32 registers)"
    return {0: "f32", 1: "bf16x3 (split-bf16 operands hi+lo, fp32 accumulate)",
            2: "f16 (vocoder: fp16 operands, 1 MFMA per product, %s; ResUNet: split-bf16 hi+lo; fp32 accumulate)" % trunk
            if gsr else "bf16x3 (split-bf16 operands hi+lo, fp32 accumulate)"}[precision]


def step_stats(step_ms, sclk_mhz):
    """Spread of the K steps of the timed region (HIP events at the step boundaries) and the median step scaled to a common
    reference clock: boxes of this pool hold 1.9 .. 2.0 GHz under this load, a 3 % spread that a single mean cannot tell from a
    3 % regression.  `step_at_ref_clock` is NOT `value` -- it is what the median step would take at 1950 MHz if time scaled with
    the shader clock alone (true for the MFMA-dense kernels, optimistic for the HBM-bound ones)."""
    if not step_ms:
        return {}
    s = sorted(step_ms)
    n = len(s)
    med = s[n // 2] if n % 2 else 0.5 * (s[n // 2 - 1] + s[n // 2])
    out = {"ms_per_step_min": round(s[0], 3), "ms_per_step_median": round(med, 3),
           "ms_per_step_p90": round(s[min(n - 1, int(np.ceil(0.9 * n)) - 1)], 3), "ms_per_step_max": round(s[-1], 3),
           "step_timing": "HIP events at the step boundaries of the timed region, on the launch stream"}
    if sclk_mhz and sclk_mhz > 500:
        out["step_at_ref_clock"] = {"ms": round(med * sclk_mhz / 1950.0, 3), "ref_sclk_mhz": 1950, "measured_avg_sclk_mhz": sclk_mhz,
                                    "definition": "median step x measured shader clock / 1950 MHz (comparison aid, never `value`)"}
    return out


def aux_workload(name, args, device, weights):
    """configs[2] / configs[4] for a few steps in the process of the default run: value, ms_per_step, parity of the benched
    batch against the FLOAT64 oracle (ssr: 2 clips, stream: the chunk), the convolution roofline (HIP events, no PMC pass)."""
    try:
        w = Workload(name, args, device, 0, 1, args.precision, weights=weights)
        steps = max(1, args.aux_steps if name == "ssr_sr64" else 10 * args.aux_steps)
        dt = w.timed(steps, 1 if name == "ssr_sr64" else 3)
        res = {"config": {"workload": name, "clips_per_gpu": w.B, "clip_seconds": w.seconds},
               "value": round(w.audio_per_step * steps / dt, 2), "unit": "audio-s/s", "steps": steps,
               "ms_per_step": round(dt / steps * 1e3, 3), "dtype": dtype_string(False, args.precision),
               "outputs_finite": bool(torch.isfinite(w.holder[0]).all().item())}
        res.update(w.extra)
        res.update(w.flags())
        got = w.holder[0].clone()
        if not args.no_roofline:
            rstep = w.step if name != "stream1s" else w.eager     # HIP events cannot be recorded inside a graph replay
            roof, whole = measure_conv_roofline(w.eng, rstep, args, None, ms_step=dt / steps * 1e3, steps=2)
            roof.pop("traffic_detail", None)
            res["roofline"], res["step"] = roof, whole
        if not args.no_parity:
            n = 2 if name == "ssr_sr64" else 1
            base, ref = cpu_baseline("ssr", w.clips, n, args.cpu_threads, repeats=args.cpu_repeats, dtype=torch.float64)
            res["cpu_baseline"] = base
            idx = sample_indices(w.B, n)
            res["parity"] = parity_wav(got[idx], ref["wav"][:, 0])
            res["parity"]["batch_indices"] = idx
            res["parity"]["vs"] = "oracle.pipeline.restore_ssr in FLOAT64 (CPU port of unet_v2.py:86-148)"
        del w
        torch.cuda.empty_cache()
        return res
    except Exception as e:  # an auxiliary line never takes the headline down with it
        return {"config": {"workload": name}, "error": repr(e)[:400]}


def aux_varlen(args, device, weights, n=128, lo=2.0, hi=8.0):
    """A test set of clips of UNEQUAL length (VCTK-shaped: 2 .. 8 s, no two alike) -- what the reference's harness iterates, one
    handler call per file (evaluation_proc/eval.py:119-134) -- through dist.restore_sharded_lengths in a world of one: sorted by
    length, one vfx_restore_gsr_varlen call per 128 clips (round 6: inside the call the ResUNet runs once per padded frame count,
    the vocoder per run of clips of similar length; round 5: one call per padded frame count).  Reported beside it: the
    same clips one call per clip (rounds 1-4), bit-equality of the two on every clip, the shortest clip against the oracle."""
    try:
        from voicefixer_main_amd import dist as vdist
        from voicefixer_main_amd import synth
        w = Workload("gsr16x10", args, device, 0, 1, args.precision, clips_n=1, seconds=1.0, weights=weights)   # (its engine)
        eng = w.eng
        rng = np.random.default_rng(2025)
        lens = [int(v) for v in rng.uniform(lo * 44100, hi * 44100, size=n)]
        base = synth.make_clips(n, hi + 0.1, seed=77)[:, 0]
        clips = [torch.from_numpy(base[i, :L].copy()).to(device) for i, L in enumerate(lens)]
        total = sum(lens) / 44100.0
        fn = vdist.checked_restore(eng)

        def timed(f, reps=3):
            f()
            torch.cuda.synchronize(device)
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                out = f()
                torch.cuda.synchronize(device)
                ts.append(time.perf_counter() - t0)
            return float(np.median(ts)), out
        dt, got = timed(lambda: vdist.restore_sharded_lengths(fn, clips, device, max_batch=128))
        dt1, one = timed(lambda: [eng.restore_gsr(c[None])[0] for c in clips], reps=1)
        res = {"config": {"workload": "varlen_vctk", "clips": n, "clip_seconds_min_max": [round(min(lens) / 44100.0, 2), round(max(lens) / 44100.0, 2)],
                          "audio_seconds": round(total, 1), "buckets": len({eng.padded_frames(L) for L in lens})},
               "value": round(total / dt, 2), "unit": "audio-s/s", "seconds": round(dt, 4), "statistic": "median of 3 passes over the set",
               "one_call_per_clip": {"value": round(total / dt1, 2), "seconds": round(dt1, 4)},
               "varlen_equals_one_call_per_clip": bool(all(torch.equal(a, b) for a, b in zip(got, one))),
               "dtype": dtype_string(True, args.precision, args.tuning)}
        res.update(w.flags())
        if not args.no_parity:
            from oracle import pipeline
            i = int(np.argmin(lens))
            unet_sd, voc_sd = synth.make_resunet_state_dict(0), synth.make_vocoder_state_dict(1)
            ref = pipeline.restore_gsr(unet_sd, voc_sd, base[i:i + 1, None, :lens[i]])
            res["parity"] = {"clip": i, "wav_sisdr_db": round(sisdr_db(got[i].cpu().numpy(), ref["wav"][0, 0]), 2),
                             "vs": "oracle.pipeline.restore_gsr on the shortest clip alone"}
        del w
        torch.cuda.empty_cache()
        return res
    except Exception as e:
        return {"config": {"workload": "varlen_vctk"}, "error": repr(e)[:400]}


# ---------------------------------------------------------------------------------------------------------
def main():
    args = parse()
    maybe_spawn(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    if args.dry_run_cpu:
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
        device = torch.device("cpu")
    else:
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        device = torch.device("cuda", local_rank)
        torch.cuda.set_device(device)

    from voicefixer_main_amd import dist as vdist

    wl = args.workload
    weights = {}
    w = Workload(wl, args, device, rank, world, args.precision, clips_n=args.clips, seconds=args.seconds, weights=weights,
                 tuning=args.tuning)
    eng, gsr, B, L = w.eng, w.gsr, w.B, w.L

    def barrier():
        if world > 1:
            dist.barrier()

    if world > 1 and args.dist_selfcheck:
        vdist.selfcheck(device)
    rccl_ranks = vdist.live_ranks(device)

    # ---- timed region ---------------------------------------------------------------------------------
    sampler = PowerSampler(int(os.environ.get("LOCAL_RANK", "0")))
    with sampler:
        dt = w.timed(args.steps, args.warmup, barrier)
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    flags = w.flags()

    res = None
    failed = None
    if rank == 0:
        res = {
            "metric": "restored-audio sec/s (RTF^-1), VoiceFixer 44.1 kHz",
            "value": round(w.audio_per_step * args.steps / dt, 2), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": dtype_string(gsr, args.precision, args.tuning),
            "data": "synthetic",
            "config": {"workload": wl, "precision_mode": args.precision, "tuning": args.tuning, "clips_per_gpu": B, "clip_seconds": w.seconds,
                       "parallelism": "dp%d" % world, "weights": "seeded random (no checkpoint available offline)"
                       + ("; built on rank 0, one flat broadcast to the other ranks" if world > 1 else "")},
            "rccl_ranks": rccl_ranks, "dry_run_cpu": bool(args.dry_run_cpu), "negative_input_flag": int(flags["negative_input"]), "f16_saturated": flags["f16_saturated"],
        }
        res.update(w.extra)
        res["power"] = sampler.result()
        res.update(step_stats(w.step_ms, (res["power"] or {}).get("avg_sclk_mhz")))
        if flags["f16_saturated"]:
            failed = ("the 16-bit vocoder clamped an activation (VFX_FLAG_F16_SATURATED): this measurement is invalid; "
                      "run with --precision 1")
        if flags["negative_input"]:
            failed = "a negative mel reached to_log (VFX_FLAG_NEGATIVE_INPUT)"
        if gsr and args.precision == 2:
            res["caveat"] = ("precision 2 runs the vocoder on fp16 operands; its waveform parity is established on seeded "
                             "synthetic vocoder weights (the pretrained TFGAN checkpoint is not obtainable offline) -- "
                             "`split_bf16_mode` is the conservative figure")

    # ---- outputs of the benched batch (for parity), oracle sample --------------------------------------
    if rank == 0 and wl == "gsr16x10":
        out_p, logmel_p = eng.restore_gsr(w.wav, want_logmel=True)
        res["outputs_finite"] = bool(torch.isfinite(out_p).all().item())
    elif rank == 0 and wl == "sharded1024":
        gathered = w.gathered
        res["outputs_finite"] = bool(torch.isfinite(gathered[0]).all().item()) and tuple(gathered[0].shape) == (w.n_total, L)
        for k in w.phases:
            res[k] = round(w.phases[k] / args.steps, 3)
        res["clips_total"] = w.n_total
        res["sub_batches_per_rank"] = -(-B // 37)        # 32-bit tensor addressing: <= 37 clips of 10 s per launch
        # the gathered result is the restore of the scattered clips: check two clips against a direct call
        chk = eng.restore_gsr(w.full[:2])
        res["gather_matches_direct_restore"] = bool(torch.equal(chk, gathered[0][:2]))
    elif rank == 0:
        res["outputs_finite"] = bool(torch.isfinite(w.holder[0]).all().item())

    if rank == 0 and world == 1 and not args.dry_run_cpu:
        traffic = None
        if not args.no_roofline:
            if args.traffic == "live" and gsr:
                traffic = live_traffic(args)
            rstep = w.step if wl != "stream1s" else w.eager      # HIP events cannot be recorded inside a graph replay
            if wl == "sharded1024":
                rstep = lambda: eng.restore_gsr(w.full[:37])     # one full sub-batch of the shard
            res["roofline"], whole = measure_conv_roofline(eng, rstep, args, traffic,
                                                           ms_step=None if wl == "sharded1024" else dt / args.steps * 1e3)
            if whole:
                res["step"] = whole
                ck = (res.get("power") or {}).get("avg_sclk_mhz")
                if ck and ck > 500:   # the same step against the MFMA rate of the clock the chip actually held under its power cap
                    whole["frac_of_mfma_peak_at_measured_sclk"] = round(whole["tflops"] / (whole["peak"] * ck / 2400.0), 4)
            res["roofline_hbm"] = measure_hbm_stages(eng, min(B, 64), L)
        if args.cpu_baseline_clips > 0:
            ssr64 = (not gsr) and not args.no_parity     # the ssr parity reference is the float64 oracle
            try:
                res["cpu_baseline"], ref = cpu_baseline("gsr" if gsr else "ssr", w.clips, args.cpu_baseline_clips, args.cpu_threads,
                                                        repeats=args.cpu_repeats, dtype=torch.float64 if ssr64 else None)
            except Exception as e:
                res["cpu_baseline"], ref = {"error": repr(e)}, None
            if ref is not None and not args.no_parity:
                n = args.cpu_baseline_clips
                idx = sample_indices(B, n)
                if wl == "gsr16x10":
                    res["parity"] = parity_gsr(out_p[idx], logmel_p[idx], ref)
                    res["parity"]["batch_indices"] = idx
                    # the clips the oracle did not see, HIP against HIP: every pair of rows (2i, 2i+1) of the benched batch,
                    # restored as its own batch of two, must equal its rows of the batch's output bit for bit (a clip's
                    # result does not depend on the batch it is in: per-clip split-K rule, no cross-clip reduction anywhere)
                    bad = [i for i in range(0, B - 1, 2)
                           if not torch.equal(eng.restore_gsr(w.wav[i:i + 2].contiguous()), out_p[i:i + 2])]
                    res["parity"]["batch_rows_equal_as_pairs"] = not bad
                    if bad:
                        failed = "restore(wav[i:i+2]) differs from rows i, i+1 of restore(wav[0:B]) for i in %s" % bad
                    if res["parity"]["logmel_l1"] > LOGMEL_L1_BAR:
                        failed = "log-mel L1 %.3g exceeds the %.0e bar" % (res["parity"]["logmel_l1"], LOGMEL_L1_BAR)
                elif wl in ("ssr_sr64", "stream1s"):
                    res["parity"] = parity_wav(w.holder[0][idx], ref["wav"][:, 0])
                    res["parity"]["batch_indices"] = idx
                    res["parity"]["vs"] = "oracle.pipeline.restore_ssr in FLOAT64 (CPU port of unet_v2.py:86-148)"
        if gsr and wl == "gsr16x10" and args.precision == 2 and not args.no_alt:
            # Same workload, same process, same box, minutes apart:
            #   split_bf16_mode -- every GEMM-shaped layer on split-bf16 operands (precision 1: 3 MFMAs per product in the
            #       vocoder as well): the stricter arithmetic, reported beside `value`, with its own parity;
            #   f32_trunk_mode  -- precision 2 with VFX_TUNE_F32_TRUNK: the round-3 data path of the vocoder (fp32 residual trunk
            #       between the ResStack launches), i.e. the A/B of round 4's fp16 trunk inside the driver's own line.
            def alt_mode(precision, tuning, dtype):
                try:
                    alt = w.make_engine(precision, tuning)
                    wav, out = w.wav, w.out
                    for _ in range(max(args.warmup, 1)):
                        alt.restore_gsr(wav, out=out)
                    torch.cuda.synchronize(device)
                    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
                    t1 = time.perf_counter()
                    ev[0].record()
                    for i in range(args.steps):
                        alt.restore_gsr(wav, out=out)
                        ev[i + 1].record()
                    torch.cuda.synchronize(device)
                    dta = time.perf_counter() - t1
                    r = {"value": round(B * w.seconds * args.steps / dta, 2), "unit": "audio-s/s",
                         "ms_per_step": round(dta / args.steps * 1e3, 3), "outputs_finite": bool(torch.isfinite(out).all().item()),
                         "dtype": dtype}
                    r.update({k: v for k, v in step_stats([ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)], None).items()
                              if k.startswith("ms_per_step")})
                    if "parity" in res:
                        sel = wav[sample_indices(B, args.cpu_baseline_clips)].contiguous()
                        o1, l1 = alt.restore_gsr(sel, want_logmel=True)
                        r["parity"] = parity_gsr(o1, l1, ref)
                    fl = alt.take_flags()
                    r["f16_saturated"] = bool(fl & 2)
                    del alt
                    return r
                except Exception as e:
                    return {"error": repr(e)}
            res["split_bf16_mode"] = alt_mode(1, 0, "bf16x3 everywhere (split-bf16 operands hi+lo, 3 MFMAs per product, fp32 accumulate)")
            if not args.tuning & F32_TRUNK:
                res["f32_trunk_mode"] = alt_mode(2, args.tuning | F32_TRUNK, dtype_string(True, 2, F32_TRUNK))
        if wl == "gsr16x10" and not args.no_aux:
            # BASELINE.json configs[2] and configs[4], driver-visible: a few steps each in this process
            del w, eng
            torch.cuda.empty_cache()
            res["aux_workloads"] = {name: aux_workload(name, args, device, weights) for name in ("ssr_sr64", "stream1s")}
            res["aux_workloads"]["varlen_vctk"] = aux_varlen(args, device, weights)
    if rank == 0:
        if failed:
            res["parity_failed"] = failed
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        sys.exit("bench.py: check failed: " + failed)


if __name__ == "__main__":
    main()
