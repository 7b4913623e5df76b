#!/usr/bin/env python3
"""First disturbed launch (round 5).  Victim: the vocoder without ResStacks (12 ops) of engine 0 on stream A; beside it on stream B
the mel ResUNet of engine 1 (itself never disturbed).  VFX_SOLO_OPS=k:99 with VFX_SOLO_PLAN_OPS=12: ops 0..k-1 of the victim run
beside the ResUNet, ops k.. with the device otherwise idle.  The smallest k with wrong results is the first disturbed op + 1."""
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
    wavs = [base[:, :100000 + 20000 * k].contiguous() for k in range(8)]
    mels = [eu.stft(w)["mel"] for w in wavs]
    ref = [ev.vocoder(m) for m in mels]
    torch.cuda.synchronize()
    os.environ["VFX_SOLO_PLAN_OPS"] = "12"
    res = {}
    detail = None
    for k in list(range(13)):
        os.environ["VFX_SOLO_OPS"] = "%d:99" % k
        bad = set()
        for _ in range(3):
            outs = []
            torch.cuda.synchronize()
            for i, m in enumerate(mels):
                with torch.cuda.stream(sb):
                    eu.resunet_mel(m)
                with torch.cuda.stream(sa):
                    outs.append(ev.vocoder(m))
            torch.cuda.synchronize()
            for i, (a, b) in enumerate(zip(outs, ref)):
                if not torch.equal(a, b):
                    bad.add(i)
                    if detail is None or detail["k"] != k:
                        d = (a - b).abs()
                        rows = torch.nonzero(d.amax(dim=1) > 0)[:, 0].tolist()
                        nz = torch.nonzero(d[rows[0]] > 0)[:, 0]
                        detail = {"k": k, "batch": i, "clips_wrong": rows, "n_diff_first_clip": int(nz.numel()),
                                  "first": int(nz[0]), "last": int(nz[-1]), "max": float(d.max()), "nan": int(torch.isnan(a).sum()),
                                  "T": int(a.shape[1])}
                        print(detail, flush=True)
        res[k] = sorted(bad)
        print("ops <%d beside the ResUNet: bad %s" % (k, sorted(bad)), flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
