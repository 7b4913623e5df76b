"""CPU oracle for the VoiceFixer 44.1 kHz inference hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement (numpy for the
DSP, torch-CPU fp32/fp64 functional ops for the convolutional stacks) of the
reference algorithm for the hot path named in BASELINE.json.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it,
and there only as the checker / the CPU baseline that is timed beside the GPU
path.  Nothing under `voicefixer_main_amd/` imports it; the product path fails
loudly when the HIP library is missing.

Pinning status:

* mel filterbank, to_log/from_log, ConvBlockRes / encoder / decoder blocks and
  both ResUNet forwards are PINNED against the reference's own modules imported
  from /root/reference (`oracle/gen_golden.py` -> `tests/golden/*.npz`).
* STFT / ISTFT restate third-party `torchlibrosa==0.0.7` (requirements.txt:10),
  which is not vendored in the reference and not installed: PARITY UNPINNED
  against torchlibrosa itself; anchored on `torch.stft/istft` (the framing the
  reference's in-repo twin `tools/dsp/base.py:52-212` uses), an explicit float64
  DFT, and the STFT->ISTFT round-trip invariant of `tools/dsp/base.py:214-232`.
* The TFGAN vocoder restates the un-pinned third-party `voicefixer` pip package
  (requirements.txt:6); neither source nor weights are available offline:
  PARITY UNPINNED (self-consistency HIP-vs-restatement only).
"""

SAMPLE_RATE = 44100
N_FFT = 2048
HOP = 441
N_BINS = N_FFT // 2 + 1
N_MELS = 128
