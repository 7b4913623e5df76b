#!/usr/bin/env python3
"""Victims and culprits across libraries (round 5).  Stream A runs a victim -- the vocoder without ResStacks (its k_voc_final
is the launch that goes wrong) or plain torch arithmetic of the same kind (a k7 conv1d to one channel + tanh; a row reduction) --
while stream B runs a culprit: this library's mel ResUNet, torch bf16 / fp32 matmuls (hipBLASLt MFMA kernels), a torch conv2d.
Every victim is compared with its own sequential result."""
import json
import os
import sys

import torch

os.environ.setdefault("VFX_NO_STREAM_TURNS", "1")      # the measurement needs the launches of the two streams to overlap

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicefixer_main_amd import synth  # noqa: E402
from voicefixer_main_amd.engine import Engine, MODEL_UNET_MEL, MODEL_VOCODER  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    usd, vsd = synth.make_resunet_state_dict(0), synth.make_vocoder_state_dict(1)
    base = torch.from_numpy(synth.make_clips(13, 7.0, seed=5)[:, 0]).to(dev)
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    ev = Engine(dev, config={"precision": 2, "tuning": 3, "voc_depth": [0] * 8})
    ev.load_state_dict(MODEL_VOCODER, vsd)
    eu = Engine(dev, config={"precision": 2})
    eu.load_state_dict(MODEL_UNET_MEL, usd)
    wavs = [base[:, :100000 + 20000 * k].contiguous() for k in range(6)]
    mels = [eu.stft(w)["mel"] for w in wavs]
    g = torch.Generator(device=dev).manual_seed(1)
    xs = [torch.randn((13, 32, 160000 + 20000 * k), device=dev, generator=g) for k in range(6)]     # (B, C, T) like the last trunk
    w7 = torch.randn((1, 32, 7), device=dev, generator=g) * 0.05
    am, bm = torch.randn((4096, 4096), device=dev, generator=g), torch.randn((4096, 4096), device=dev, generator=g)
    ah, bh = am.bfloat16(), bm.bfloat16()
    xc = torch.randn((13, 32, 256, 128), device=dev, generator=g)
    wc = torch.randn((32, 32, 3, 3), device=dev, generator=g) * 0.05
    torch.cuda.synchronize()
    victims = {
        "vfx vocoder (k_voc_final)": lambda i: ev.vocoder(mels[i]),
        "torch conv1d k7 + tanh": lambda i: torch.tanh(torch.nn.functional.conv1d(torch.nn.functional.leaky_relu(xs[i], 0.2), w7, padding=3)),
        "torch row sum": lambda i: xs[i].sum(dim=1),
    }
    culprits = {
        "vfx mel ResUNet": lambda i: eu.resunet_mel(mels[i]),
        "torch bf16 matmul x8": lambda i: [torch.mm(ah, bh) for _ in range(8)],
        "torch fp32 matmul x4": lambda i: [torch.mm(am, bm) for _ in range(4)],
        "torch conv2d 3x3 x4": lambda i: [torch.nn.functional.conv2d(xc, wc, padding=1) for _ in range(4)],
    }
    res = {}
    for vn, vic in victims.items():
        ref = [vic(i) for i in range(6)]
        torch.cuda.synchronize()
        for cn, cul in culprits.items():
            cul(0)
            torch.cuda.synchronize()
            bad, nwrong = set(), 0
            for _ in range(4):
                outs = []
                torch.cuda.synchronize()
                for i in range(6):
                    with torch.cuda.stream(sb):
                        cul(i)
                    with torch.cuda.stream(sa):
                        outs.append(vic(i))
                torch.cuda.synchronize()
                for i, (a, b) in enumerate(zip(outs, ref)):
                    if not torch.equal(a, b):
                        bad.add(i)
                        nwrong += int((a != b).sum())
            res["%s beside %s" % (vn, cn)] = {"bad_batches": sorted(bad), "wrong_values": nwrong}
            print("%s beside %s: bad %s, %d wrong values" % (vn, cn, sorted(bad), nwrong), flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
