"""CPU restatement of the handlers' spectral metrics (TEST INFRASTRUCTURE ONLY; see oracle/__init__.py).

evaluation_proc/metrics.py:83-95 (`AudioMetrics.lsd`, `AudioMetrics.sispec`, EPS = 1e-12 :16) and the helpers they call,
evaluation_proc/utils.py:81-101 (`pow_p_norm`, `energy_unify`, `pow_norm`, EPS = 1e-8 :8).  float64 numpy, per clip.
Pinned by tests/golden/metrics.npz, which oracle/gen_golden_metrics.py produces by executing the reference's own
function bodies (extracted from the two files; the modules themselves need packages that are not installed).
"""
import numpy as np

EPS_METRICS = 1e-12   # evaluation_proc/metrics.py:16
EPS_UTILS = 1e-8      # evaluation_proc/utils.py:8


def lsd(est, target):
    """metrics.py:83-87: (B, C, T, F) linear -> (B, C) = mean_t sqrt(mean_f log10(tgt^2 / (est + eps)^2 + eps)^2)."""
    est, target = np.asarray(est, np.float64), np.asarray(target, np.float64)
    v = np.log10(target ** 2 / (est + EPS_METRICS) ** 2 + EPS_METRICS) ** 2
    return np.mean(np.mean(v, axis=3) ** 0.5, axis=2)


def sispec_per_clip(est, target):
    """metrics.py:89-95 with utils.py:81-101, before the batch mean: (B, C, T, F) -> (B,) in dB."""
    est, target = np.asarray(est, np.float64), np.asarray(target, np.float64)
    B = est.shape[0]
    scale = np.sum(est * target, axis=tuple(range(2, est.ndim)), keepdims=True)          # pow_norm
    pp = np.sum(target.reshape(B, -1) ** 2, axis=1).reshape((B,) + (1,) * (est.ndim - 1))   # pow_p_norm
    tgt = scale * target / (pp + EPS_UTILS)                                                 # energy_unify
    noise = est - tgt
    pt = np.sum(tgt.reshape(B, -1) ** 2, axis=1)
    pn = np.sum(noise.reshape(B, -1) ** 2, axis=1)
    return 10 * np.log10(pt / (pn + EPS_METRICS) + EPS_METRICS)


def sispec(est, target):
    """The reference's return value: the batch mean."""
    v = sispec_per_clip(est, target)
    return float(np.sum(v) / v.shape[0])
