import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def engine():
    """One libvfx handle with the synthetic mel-ResUNet + vocoder weights loaded (GPU tests)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from voicefixer_main_amd import synth
    from voicefixer_main_amd.engine import Engine, MODEL_UNET_MEL, MODEL_VOCODER
    eng = Engine("cuda:0")
    eng.load_state_dict(MODEL_UNET_MEL, synth.make_resunet_state_dict(0))
    eng.load_state_dict(MODEL_VOCODER, synth.make_vocoder_state_dict(1))
    return eng


@pytest.fixture(scope="session")
def unet_sd():
    from voicefixer_main_amd import synth
    return synth.make_resunet_state_dict(0)


@pytest.fixture(scope="session")
def voc_sd():
    from voicefixer_main_amd import synth
    return synth.make_vocoder_state_dict(1)
