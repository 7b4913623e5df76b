#!/usr/bin/env python3
"""Round 6, the closing measurement of the two-stream finding (profiles/r05_two_streams.md): is the disturbance chip-wide or
XCD-local, and does the hardware report anything?

Victim (stream A): the vocoder without ResStacks -- its k_voc_final is the launch that goes wrong.  Aggressor (stream B): this
library's mel ResUNet (split-bf16 MFMA convolutions).  The two streams are created with CU masks (hipExtStreamCreateWithCUMask);
`vfx_debug_where` (libvfx_test.so) reads XCC_ID / HW_ID on each stream first, so every row says where its kernels really ran.
  plain        two ordinary streams (the round-5 condition)
  xcd_split    victim on XCDs 0-3, aggressor on XCDs 4-7 (no shared CU, no shared L2)
  cu_split     both on all eight XCDs, victim on the even CUs of the mask order, aggressor on the odd ones (shared L2s, no shared CU)
  same_half    both on XCDs 0-3 (control: masks as such do not cure anything)
  two_procs    the aggressor in a SECOND PROCESS on the same GPU, one stream each
RAS error counters (sysfs + rocm-smi --showrasinfo) are read before and after."""
import ctypes
import glob
import json
import os
import subprocess
import sys
import time

import torch

os.environ.setdefault("VFX_NO_STREAM_TURNS", "1")      # the measurement needs the launches of the two streams to overlap

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicefixer_main_amd import _lib, synth  # noqa: E402
from voicefixer_main_amd.engine import Engine, MODEL_UNET_MEL, MODEL_VOCODER  # noqa: E402

FLAG = "/tmp/vfx_two_procs_ready"


def ras_snapshot():
    out = {}
    for f in sorted(glob.glob("/sys/class/drm/card*/device/ras/*_err_count")):
        try:
            out[f.split("/device/ras/")[1]] = open(f).read().strip().replace("\n", " ")
        except OSError as e:
            out[f] = "unreadable: %s" % e
    try:
        r = subprocess.run(["rocm-smi", "--showrasinfo", "all"], capture_output=True, text=True, timeout=60)
        out["rocm-smi"] = [l for l in r.stdout.splitlines() if l.strip() and "====" not in l][:40]
    except Exception as e:  # noqa: BLE001
        out["rocm-smi"] = "failed: %s" % e
    return out


def masked_stream(hip, words):
    arr = (ctypes.c_uint32 * len(words))(*words)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), arr)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask -> %d" % rc)
    return torch.cuda.ExternalStream(s.value)


def where(stream, tlib, dev):
    """-> {xcc: {(se, sh, cu), ...}} of a 4096-block probe launch on `stream`."""
    out = torch.zeros(2048, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    tlib.vfx_debug_where(ctypes.c_void_p(out.data_ptr()), 4096, 64, ctypes.c_void_p(stream.cuda_stream))
    torch.cuda.synchronize()
    h = out.cpu().tolist()
    res = {}
    for k, n in enumerate(h):
        if n:
            res.setdefault(k >> 8, set()).add(k & 255)
    return res


def bits(pred, n=256):
    words = [0] * (n // 32)
    for i in range(n):
        if pred(i):
            words[i // 32] |= 1 << (i % 32)
    return words


def aggressor_loop(seconds):
    dev = torch.device("cuda:0")
    eu = Engine(dev, config={"precision": 2})
    eu.load_state_dict(MODEL_UNET_MEL, synth.make_resunet_state_dict(0))
    base = torch.from_numpy(synth.make_clips(13, 7.0, seed=5)[:, 0]).to(dev)
    mel = eu.stft(base[:, :200000].contiguous())["mel"]
    eu.resunet_mel(mel)
    torch.cuda.synchronize()
    open(FLAG, "w").write("ready")
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        for _ in range(4):
            eu.resunet_mel(mel)
        torch.cuda.synchronize()
        n += 4
    print("aggressor process: %d ResUNet calls" % n, flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--aggressor":
        aggressor_loop(float(sys.argv[2]))
        return
    dev = torch.device("cuda:0")
    hip = ctypes.CDLL("libamdhip64.so")
    tlib = _lib.load_test()
    tlib.vfx_debug_where.restype = ctypes.c_int
    tlib.vfx_debug_where.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    res = {"ras_before": ras_snapshot()}
    usd, vsd = synth.make_resunet_state_dict(0), synth.make_vocoder_state_dict(1)
    base = torch.from_numpy(synth.make_clips(13, 7.0, seed=5)[:, 0]).to(dev)
    ev = Engine(dev, config={"precision": 2, "tuning": 3, "voc_depth": [0] * 8})
    ev.load_state_dict(MODEL_VOCODER, vsd)
    eu = Engine(dev, config={"precision": 2})
    eu.load_state_dict(MODEL_UNET_MEL, usd)
    wavs = [base[:, :100000 + 20000 * k].contiguous() for k in range(6)]
    mels = [eu.stft(w)["mel"] for w in wavs]
    ref = [ev.vocoder(m) for m in mels]
    for m in mels:
        eu.resunet_mel(m)
    torch.cuda.synchronize()

    # which mask bit is which XCD?  probe single-bit-class masks
    probe = {}
    for name, pred in (("bits i%8<4", lambda i: i % 8 < 4), ("bits i<128", lambda i: i < 128)):
        try:
            s = masked_stream(hip, bits(pred))
            probe[name] = {x: len(c) for x, c in where(s, tlib, dev).items()}
        except Exception as e:  # noqa: BLE001
            probe[name] = "failed: %s" % e
    res["mask_probe_cus_per_xcc"] = probe
    print("mask probe:", probe, flush=True)
    interleaved = isinstance(probe.get("bits i%8<4"), dict) and set(probe["bits i%8<4"]) <= {0, 1, 2, 3}
    contiguous = isinstance(probe.get("bits i<128"), dict) and set(probe["bits i<128"]) <= {0, 1, 2, 3}
    xcd_of = (lambda i: i % 8) if interleaved else ((lambda i: i // 32) if contiguous else None)
    cu_of = (lambda i: i // 8) if interleaved else (lambda i: i % 32)
    res["mask_layout"] = "interleaved (bit i -> XCD i % 8)" if interleaved else ("contiguous (bit i -> XCD i // 32)" if contiguous else "unknown")

    def run(tag, sa, sb):
        wa = {x: len(c) for x, c in where(sa, tlib, dev).items()}
        wb = {x: len(c) for x, c in where(sb, tlib, dev).items()}
        bad, nwrong, nb = set(), 0, 0
        for _ in range(6):
            outs = []
            torch.cuda.synchronize()
            for i in range(6):
                with torch.cuda.stream(sb):
                    eu.resunet_mel(mels[i])
                with torch.cuda.stream(sa):
                    outs.append(ev.vocoder(mels[i]))
            torch.cuda.synchronize()
            for i, (a, b) in enumerate(zip(outs, ref)):
                nb += 1
                if not torch.equal(a, b):
                    bad.add(i)
                    nwrong += int((a != b).sum())
        res[tag] = {"victim_cus_per_xcc": wa, "aggressor_cus_per_xcc": wb, "batches": nb, "bad_batch_ids": sorted(bad), "wrong_values": nwrong}
        print(tag, res[tag], flush=True)

    run("plain", torch.cuda.Stream(dev), torch.cuda.Stream(dev))
    if xcd_of is not None:
        run("xcd_split", masked_stream(hip, bits(lambda i: xcd_of(i) < 4)), masked_stream(hip, bits(lambda i: xcd_of(i) >= 4)))
        run("cu_split", masked_stream(hip, bits(lambda i: cu_of(i) % 2 == 0)), masked_stream(hip, bits(lambda i: cu_of(i) % 2 == 1)))
        run("same_half", masked_stream(hip, bits(lambda i: xcd_of(i) < 4)), masked_stream(hip, bits(lambda i: xcd_of(i) < 4)))
    run("plain_again", torch.cuda.Stream(dev), torch.cuda.Stream(dev))

    # the aggressor in a second process
    if os.path.exists(FLAG):
        os.remove(FLAG)
    child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--aggressor", "25"])
    t0 = time.time()
    while not os.path.exists(FLAG) and time.time() - t0 < 180 and child.poll() is None:
        time.sleep(0.5)
    bad, nwrong, nb = set(), 0, 0
    t1 = time.time()
    while time.time() - t1 < 15 and child.poll() is None:
        outs = [ev.vocoder(mels[i]) for i in range(6)]
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip(outs, ref)):
            nb += 1
            if not torch.equal(a, b):
                bad.add(i)
                nwrong += int((a != b).sum())
    child.wait(timeout=120)
    res["two_procs"] = {"batches": nb, "bad_batch_ids": sorted(bad), "wrong_values": nwrong, "child_rc": child.returncode}
    print("two_procs", res["two_procs"], flush=True)
    res["ras_after"] = ras_snapshot()
    res["ras_changed"] = res["ras_after"] != res["ras_before"]
    print(json.dumps(res, default=lambda o: sorted(o) if isinstance(o, set) else str(o)))


if __name__ == "__main__":
    main()
