import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Tolerances per arithmetic mode.  The north-star bar is log-mel L1 <= 1e-3.
TOL = {
    0: dict(name="fp32", logmel_l1=2e-5, logmel_max=5e-4, conv=2e-5, voc_max=2e-5, sisdr=80.0),
    1: dict(name="split-bf16", logmel_l1=2e-4, logmel_max=3e-3, conv=3e-4, voc_max=1e-4, sisdr=60.0),
    # precision 2: the ResUNets as 1; the vocoder (and the single-op entry points) on fp16 operands, 1 MFMA per product
    2: dict(name="fp16-vocoder", logmel_l1=2e-4, logmel_max=3e-3, conv=3e-3, voc_max=1e-3, sisdr=52.0),
}


def _make_engine(precision):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from voicefixer_main_amd import synth
    from voicefixer_main_amd.engine import Engine, MODEL_UNET_MEL, MODEL_VOCODER
    eng = Engine("cuda:0", config={"precision": precision})
    eng.load_state_dict(MODEL_UNET_MEL, synth.make_resunet_state_dict(0))
    eng.load_state_dict(MODEL_VOCODER, synth.make_vocoder_state_dict(1))
    eng.tol = TOL[precision]
    return eng


@pytest.fixture(scope="session", params=[1, 0, 2], ids=["split-bf16", "fp32", "fp16-vocoder"])
def engine(request):
    """libvfx handle with the synthetic mel-ResUNet + vocoder weights, in each arithmetic mode."""
    return _make_engine(request.param)


@pytest.fixture(scope="session")
def unet_sd():
    from voicefixer_main_amd import synth
    return synth.make_resunet_state_dict(0)


@pytest.fixture(scope="session")
def voc_sd():
    from voicefixer_main_amd import synth
    return synth.make_vocoder_state_dict(1)
