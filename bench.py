#!/usr/bin/env python3
"""bench.py -- VoiceFixer 44.1 kHz restoration throughput on MI355X.

Metric (BASELINE.json): restored-audio seconds per wall-second (RTF^-1).  A "step" is one
pass of the whole `gsr_voicefixer` hot path (STFT -> mel -> ResUNet -> from_log -> TFGAN
vocoder -> peak normalise -> trim; eval_gsr_voicefixer.py:47-74) over one batch of
16 x 10 s synthetic clips per GPU (BASELINE.json configs[1]), inputs resident in HBM.
Weak scaling: every rank restores its own shard, no collective on the data path.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Arithmetic (--precision): 2 (default) = BASELINE.json's 16-bit operand mode for
configs[1]: the ResUNet on split-bf16 operands (it carries the log-mel L1 <= 1e-3 bar), the vocoder on
fp16 operands with one MFMA per product; 1 = split-bf16 everywhere (3 MFMAs per product), timed as well
at N = 1 and reported beside `value` as `split_bf16_mode`; 0 = exact fp32 MFMA.

Rank 0 prints ONE JSON line.  `roofline` is measured live: HIP events around every convolution
launch on its own stream over K more steps of the same workload, reported for the kernel with the
largest share of GPU time; `cpu_baseline` times the CPU oracle (oracle/, a port of the reference
algorithm) on a bounded sample of the same clips on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (spec; 2495 measured)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--clips", type=int, default=16, help="clips per GPU per step")
    ap.add_argument("--seconds", type=float, default=10.0, help="clip length")
    ap.add_argument("--cpu-baseline-clips", type=int, default=2, help="clips in the CPU-oracle sample (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=32)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dist-selfcheck", action="store_true", help="N > 1: round-trip a tensor through dist.scatter_clips / gather_clips first")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra all-split-bf16 (precision 1) timing at N = 1")
    ap.add_argument("--precision", type=int, default=2, help="0 = exact fp32 MFMA, 1 = split-bf16 (hi+lo, 3 bf16 MFMAs), 2 = ResUNet split-bf16 + vocoder fp16 (1 MFMA per product)")
    return ap.parse_args()


def cpu_baseline(clips, n_clips, threads):
    """Time the CPU oracle on a bounded sample of the same workload (rank 0, N=1 only).

    The oracle's torch-CPU convolutions stop scaling (and then collapse) beyond a few dozen
    threads at these sizes (measured on the 2 x EPYC 9575F host: 1-s clip 0.23 / 0.18 / 0.25 /
    0.66 s at 8 / 16 / 32 / 64 threads, minutes at 256), so the baseline uses a fixed, stated
    thread count instead of every hardware thread.
    """
    from oracle import pipeline
    from voicefixer_main_amd import synth
    wav = clips[:n_clips]
    unet_sd = synth.make_resunet_state_dict(0)
    voc_sd = synth.make_vocoder_state_dict(1)
    threads = max(1, min(threads, os.cpu_count() or 1))
    torch.set_num_threads(threads)
    pipeline.restore_gsr(unet_sd, voc_sd, wav[:1, :, :44100])          # warm-up (thread pool, allocator)
    t0 = time.perf_counter()
    pipeline.restore_gsr(unet_sd, voc_sd, wav)
    dt = time.perf_counter() - t0
    seconds = wav.shape[0] * wav.shape[-1] / 44100.0
    return {"value": round(seconds / dt, 3), "unit": "audio-s/s", "cores": threads, "kind": "port",
            "seconds": round(dt, 2), "host_cpus": os.cpu_count(),
            "sample": "oracle.pipeline.restore_gsr (torch-CPU fp32 + numpy port of the reference algorithm) on "
                      "%d clip(s) x %.0f s of the same synthetic clips as one batch, same seeded weights, after a "
                      "1-s warm-up" % (wav.shape[0], wav.shape[-1] / 44100.0)}


def measure_roofline(eng, wav, out, args):
    """HIP events around every convolution launch (on the launch stream) over K more steps of the same
    workload; per-kernel totals come from the per-launch table libvfx writes (VFX_PROFILE_DUMP).  The
    roofline object describes the kernel with the largest share of GPU time; `traffic` is the HBM byte count
    of that kernel per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes, + WRITE_SIZE), null if no such measurement exists for this workload."""
    import csv
    import tempfile
    keep = os.environ.get("VFX_PROFILE_DUMP")   # a caller-provided path keeps the per-launch table
    dump = keep or tempfile.NamedTemporaryFile(prefix="vfx_convs_", suffix=".csv", delete=False).name
    os.environ["VFX_PROFILE_DUMP"] = dump
    eng.profile_begin()
    for _ in range(args.steps):
        eng.restore_gsr(wav, out=out)
    n, ms, fl = eng.profile_end()
    rows = list(csv.DictReader(open(dump)))
    if not keep:
        os.environ.pop("VFX_PROFILE_DUMP", None)
        os.unlink(dump)
    per = {}
    for r in rows:
        k = r["kernel"].replace(";", ",")
        t = per.setdefault(k, [0, 0.0, 0.0])
        t[0] += 1
        t[1] += float(r["ms"])
        t[2] += float(r["tflops"]) * float(r["ms"]) * 1e9   # flops of the launch
    steps = max(args.steps, 1)
    dom = max(per, key=lambda k: per[k][1])
    cnt, kms, kfl = per[dom]
    tflops = kfl / (kms * 1e-3) / 1e12
    split = args.precision >= 1
    peak = PEAK_BF16_MFMA_TFLOPS if split else PEAK_FP32_MFMA_TFLOPS
    traffic, traffic_detail = None, None
    tpath = os.path.join(HERE, "profiles", "r01_traffic.json")
    if os.path.exists(tpath):
        try:
            t = json.load(open(tpath))
            if t.get("workload") == "%dx%.0fs" % (wav.shape[0], args.seconds) and t.get("precision") == args.precision:
                traffic_detail = t.get("kernels", {}).get(dom)
                traffic = traffic_detail["bytes_per_launch"] if traffic_detail else None
        except Exception:
            traffic, traffic_detail = None, None
    plain = "f16" in dom.split(">")[-1]      # a 16-bit launch of the precision-2 vocoder: one MFMA per product
    per_product = 1 if (plain or not split) else 3
    return {
        "bound": "mfma",
        "kernel": "%s (%s)" % (dom, "1 x v_mfma_f32_32x32x16_f16 per product (fp16 operands), fp32 accumulate" if plain
                               else "3 x v_mfma_f32_32x32x16_bf16 per product (hi*hi + hi*lo + lo*hi), fp32 accumulate"
                               if split else "v_mfma_f32_32x32x2_f32"),
        "achieved": round(tflops, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tflops / peak, 4),
        # achieved counts ALGORITHMIC flops (2*M*N*K once); in split-bf16 mode the kernel issues 3 bf16 MFMAs per
        # product, so `frac` is bounded by 1/3 and mfma_issue_frac is the share of the MFMA pipe actually used
        "mfma_issue_frac": round(tflops * per_product / peak, 4),
        "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC passes of profiles/r01_traffic.json)",
        "traffic_detail": traffic_detail,
        "launches_per_step": cnt // steps,
        "avg_launch_us": round(kms * 1e3 / max(cnt, 1), 2),
        "kernel_ms_per_step": round(kms / steps, 3),
        "algorithmic_gflop_per_step": round(kfl / steps / 1e9, 1),
        "share_of_conv_time": round(kms / max(ms, 1e-9), 4),
        "all_conv_kernels": {k: {"launches_per_step": v[0] // steps, "ms_per_step": round(v[1] / steps, 3),
                                 "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 1)} for k, v in sorted(per.items())},
        "all_conv_ms_per_step": round(ms / steps, 3),
        "all_conv_algorithmic_gflop_per_step": round(fl / steps / 1e9, 1),
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from voicefixer_main_amd import synth
    from voicefixer_main_amd.engine import Engine, MODEL_UNET_MEL, MODEL_VOCODER

    eng = Engine(device, config={"precision": args.precision})
    eng.load_state_dict(MODEL_UNET_MEL, synth.make_resunet_state_dict(0))
    eng.load_state_dict(MODEL_VOCODER, synth.make_vocoder_state_dict(1))

    clips = synth.make_clips(args.clips, args.seconds, seed=1234 + 1000 * rank)      # (B, 1, L) float32, host
    wav = torch.from_numpy(clips[:, 0]).to(device)                                   # resident in HBM
    out = torch.empty_like(wav)
    B, L = wav.shape

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    if world > 1 and args.dist_selfcheck:
        # Exercise the shard scatter / gather over RCCL once, outside the timed region (opt-in: a point-to-point
        # problem on one rank would otherwise hang the whole measurement; the logic itself is covered on gloo).
        try:
            from voicefixer_main_amd import dist as vdist
            vdist.selfcheck(device)
        except Exception as e:  # never fatal for the throughput measurement
            print("[bench] scatter/gather self-check failed on rank %d: %r" % (rank, e), file=sys.stderr)

    for _ in range(args.warmup):
        eng.restore_gsr(wav, out=out)
    torch.cuda.synchronize(device)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.restore_gsr(wav, out=out)
    torch.cuda.synchronize(device)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    flags = eng.take_flags()
    finite = bool(torch.isfinite(out).all().item())

    roofline = None
    if not args.no_roofline and rank == 0:
        roofline = measure_roofline(eng, wav, out, args)

    if rank == 0:
        audio_s = world * B * args.seconds * args.steps
        res = {
            "metric": "restored-audio sec/s (RTF^-1), VoiceFixer 44.1 kHz",
            "value": round(audio_s / dt, 2), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": {0: "f32", 1: "bf16x3 (split-bf16 operands hi+lo, fp32 accumulate)",
                      2: "f16 (vocoder: fp16 operands, 1 MFMA per product; ResUNet: split-bf16 hi+lo; fp32 accumulate)"}[args.precision],
            "data": "synthetic",
            "config": {"workload": "gsr_voicefixer ResUNet+vocoder restore, batch=%dx%.0f s @44.1 kHz per GPU "
                                   "(BASELINE.json configs[1], 16-bit operands); ResUNet: split-bf16 (plain bf16 misses the "
                                   "log-mel L1<=1e-3 bar 19x, split meets it with >10x margin: 4e-5); vocoder: fp16 operands, "
                                   "waveform SI-SDR 58 dB and restored-waveform log-mel L1 2.4e-4 vs the fp32 oracle"
                                   % (B, args.seconds),
                       "precision_mode": args.precision,
                       "clips_per_gpu": B, "clip_seconds": args.seconds, "parallelism": "dp%d" % world,
                       "weights": "seeded random (no checkpoint available offline)"},
            "outputs_finite": finite, "negative_input_flag": flags,
        }
        if roofline:
            res["roofline"] = roofline
        if world == 1 and args.precision == 2 and not args.no_alt:
            # Same workload with every GEMM-shaped layer on split-bf16 operands (precision 1: 3 MFMAs per product in the
            # vocoder as well): the stricter arithmetic, reported beside `value`.
            try:
                alt = Engine(device, config={"precision": 1})
                alt.load_state_dict(MODEL_UNET_MEL, synth.make_resunet_state_dict(0))
                alt.load_state_dict(MODEL_VOCODER, synth.make_vocoder_state_dict(1))
                for _ in range(max(args.warmup, 1)):
                    alt.restore_gsr(wav, out=out)
                torch.cuda.synchronize(device)
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    alt.restore_gsr(wav, out=out)
                torch.cuda.synchronize(device)
                dta = time.perf_counter() - t1
                res["split_bf16_mode"] = {
                    "value": round(B * args.seconds * args.steps / dta, 2), "unit": "audio-s/s",
                    "ms_per_step": round(dta / args.steps * 1e3, 3), "outputs_finite": bool(torch.isfinite(out).all().item()),
                    "dtype": "bf16x3 everywhere (split-bf16 operands hi+lo, 3 MFMAs per product, fp32 accumulate)",
                    "parity": "waveform SI-SDR 88-94 dB vs the fp32 oracle; value's mode: 58 dB "
                              "(profiles/r01_parity_fp16_vocoder.json)"}
                del alt
            except Exception as e:
                res["split_bf16_mode"] = {"error": repr(e)}
        if world == 1 and args.cpu_baseline_clips > 0:
            try:
                res["cpu_baseline"] = cpu_baseline(clips, args.cpu_baseline_clips, args.cpu_threads)
            except Exception as e:
                res["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(res))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
