#!/usr/bin/env python3
"""tests/golden/metrics.npz from the REFERENCE's own metric code (build container only).

    python oracle/gen_golden_metrics.py

evaluation_proc/metrics.py and evaluation_proc/utils.py cannot be imported here (git, librosa, skimage, speechmetrics
are not installed), so the bodies of `AudioMetrics.lsd`, `AudioMetrics.sispec`, `pow_p_norm`, `energy_unify` and
`pow_norm` are cut out of the two files with `ast` and executed unmodified, with the module-level EPS each file defines.
Test infrastructure only.
"""
import ast
import os
import sys
import textwrap

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def cut(path, names):
    src = open(path).read()
    tree = ast.parse(src)
    out = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in names:
            out[node.name] = textwrap.dedent(ast.get_source_segment(src, node))
    eps = [n for n in tree.body if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "EPS"]
    return out, float(ast.literal_eval(eps[0].value))


def main():
    ufn, ueps = cut(os.path.join(REF, "evaluation_proc", "utils.py"), {"pow_p_norm", "energy_unify", "pow_norm"})
    mfn, meps = cut(os.path.join(REF, "evaluation_proc", "metrics.py"), {"lsd", "sispec"})
    uns = {"torch": torch, "EPS": ueps}
    for name in ("pow_p_norm", "pow_norm", "energy_unify"):
        exec(ufn[name], uns)
    mns = {"torch": torch, "EPS": meps, "energy_unify": uns["energy_unify"], "pow_p_norm": uns["pow_p_norm"]}
    for name in ("lsd", "sispec"):
        exec(mfn[name], mns)
    rng = np.random.RandomState(7)
    out = {"eps_metrics": meps, "eps_utils": ueps}
    for tag, B, T, F, noise in (("a", 2, 37, 128, 0.3), ("b", 3, 11, 1025, 0.02), ("c", 1, 5, 128, 1e-4)):
        tgt = (10.0 ** (rng.normal(size=(B, 1, T, F)) * 0.8 - 1.0)).astype(np.float32)
        est = (tgt * (1.0 + noise * rng.normal(size=tgt.shape))).clip(0, None).astype(np.float32)
        te, tt = torch.from_numpy(est), torch.from_numpy(tgt)
        out[tag + "_est"], out[tag + "_tgt"] = est, tgt
        out[tag + "_lsd"] = mns["lsd"](None, te, tt).numpy()
        out[tag + "_sispec_lin"] = np.float64(mns["sispec"](None, te.clone(), tt.clone()))
        le, lt = torch.log10(te.clamp(min=1e-8)), torch.log10(tt.clamp(min=1e-8))
        out[tag + "_sispec_log"] = np.float64(mns["sispec"](None, le, lt))
    path = os.path.join(ROOT, "tests", "golden", "metrics.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if "est" not in k and "tgt" not in k})


if __name__ == "__main__":
    main()
