"""ctypes binding of libvfx.so (the C ABI declared in include/vfx.h) and, for the tests, libvfx_test.so (include/vfx_test.h).

The product path has NO fallback: if the shared library cannot be loaded, or a call
fails, a RuntimeError is raised.  PyTorch is used only for device memory and streams.
"""
import ctypes
import os
import subprocess
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VFX_LIB_PATH") or os.path.join(HERE, "libvfx.so")   # VFX_LIB_PATH: ablation builds (scripts/)
TEST_LIB_PATH = os.path.join(HERE, "libvfx_test.so")
CSRC = os.path.join(HERE, "csrc")

VFX_MAX_STAGES = 8
MODEL_UNET_MEL, MODEL_UNET_SPEC, MODEL_VOCODER, MODEL_FRONTEND = 0, 1, 2, 3
# vfx_config.tuning bits (include/vfx.h)
TUNE_NO_FUSED_STACKS, TUNE_NO_FUSED_WIDE, TUNE_NO_FUSED_UNET, TUNE_NO_PERSISTENT_C64, TUNE_NO_PAIRS, TUNE_NO_SPLITK, \
    TUNE_F32_TRUNK, TUNE_SMALL_2D_TILES, TUNE_DEBUG_POISON_ARENA, TUNE_NO_FUSED_UPSAMPLERS, TUNE_OLD_BLOCK2D, \
    TUNE_TWO_LAUNCH_UPSAMPLERS = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048
FLAG_NEGATIVE_INPUT = 1
FLAG_F16_SATURATED = 2   # precision 2: an activation of the vocoder left the fp16 range and was clamped
FLAG_PEAK_NORMALISED = 4  # vfx_restore_gsr divided a clip by its peak (the reference's "Exceed energy limit" warning)


class VfxConfig(ctypes.Structure):
    _fields_ = [
        ("sample_rate", c_int), ("n_fft", c_int), ("hop", c_int), ("n_mels", c_int),
        ("voc_cond_channels", c_int), ("voc_cond_layers", c_int), ("voc_channels", c_int),
        ("voc_n_stages", c_int), ("voc_scales", c_int * VFX_MAX_STAGES), ("voc_depth", c_int * VFX_MAX_STAGES),
        ("voc_dilation_base", c_int), ("voc_min_db", c_float), ("voc_amp_floor", c_float),
        ("voc_norm_range", c_float), ("voc_up_slope", c_float), ("voc_res_slope", c_float),
        ("precision", c_int), ("tuning", c_int),
    ]


# name -> (restype, argtypes): every symbol include/vfx.h declares
SIGNATURES = {
    "vfx_default_config": (c_int, [POINTER(VfxConfig)]),
    "vfx_create": (c_int, [c_int, POINTER(VfxConfig), POINTER(c_void_p)]),
    "vfx_destroy": (c_int, [c_void_p]),
    "vfx_last_error": (c_char_p, []),
    "vfx_load_tensor": (c_int, [c_void_p, c_int, c_char_p, c_void_p, POINTER(c_int64), c_int]),
    "vfx_finalize_weights": (c_int, [c_void_p, c_int]),
    "vfx_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int]),
    "vfx_reserve": (c_int, [c_void_p, c_int, c_int, c_int]),
    "vfx_unpin_plans": (c_int, [c_void_p]),
    "vfx_stft_mel": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "vfx_stft_phase": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p]),
    "vfx_mel_project": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "vfx_istft": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vfx_spectral_metrics": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vfx_chunk_gather": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vfx_chunk_ola": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_int, c_int, c_int,
                      c_void_p, c_void_p]),
    "vfx_resunet_mel": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "vfx_resunet_spec": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vfx_vocoder_out_len": (c_int64, [c_void_p, c_int]),
    "vfx_vocoder": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "vfx_restore_gsr": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "vfx_restore_gsr_varlen": (c_int, [c_void_p, c_void_p, c_int, c_int, POINTER(c_int), c_void_p, c_void_p, c_int, c_void_p]),
    "vfx_restore_ssr_varlen": (c_int, [c_void_p, c_void_p, c_int, c_int, POINTER(c_int), c_void_p, c_void_p]),
    "vfx_take_flags": (c_int, [c_void_p, c_void_p, POINTER(c_int)]),
    "vfx_take_flags_masked": (c_int, [c_void_p, c_void_p, c_int, POINTER(c_int)]),
    "vfx_turn_begin": (c_int, [c_int, c_void_p]),
    "vfx_turn_end": (c_int, [c_int, c_void_p]),
    "vfx_profile_begin": (c_int, [c_void_p]),
    "vfx_profile_end": (c_int, [c_void_p, POINTER(c_int64), POINTER(ctypes.c_double), POINTER(ctypes.c_double)]),
}

# name -> (restype, argtypes): every symbol include/vfx_test.h declares (libvfx_test.so: kernel-level entry points of the parity
# tests; the product path never loads it)
TEST_SIGNATURES = {
    "vfx_op_conv": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                            c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vfx_op_conv_transpose": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int,
                                      c_int, c_int, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "vfx_op_resblock": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                c_float, c_int, c_void_p, c_void_p]),
    "vfx_plan_resblock_geometry": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vfx_plan_resblock_geometry_tuned": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vfx_plan_block2d_geometry": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vfx_plan_conv_geometry": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "vfx_op_resblock_pair": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p]),
    "vfx_op_block2d": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
}

_lib = None
_test_lib = None


def build(verbose=False):
    """Compile libvfx.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    out = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout[-4000:])
        print(out.stderr[-4000:])
    if out.returncode != 0:
        raise RuntimeError("building libvfx.so failed")
    return LIB_PATH


def load():
    """Load libvfx.so; raises RuntimeError (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libvfx.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the product path)" % LIB_PATH)
    try:
        # default RTLD_LOCAL: the product path does not put the library's internals into the process-wide namespace beside
        # torch's and HIP's symbols (load_test promotes the already-loaded copy when the test library needs them)
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 missing
        raise RuntimeError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def load_test():
    """Load libvfx_test.so (kernel-level entry points of the parity tests) on top of libvfx.so; raises when it is missing."""
    global _test_lib
    if _test_lib is not None:
        return _test_lib
    load()
    if not os.path.exists(TEST_LIB_PATH):
        raise RuntimeError("libvfx_test.so not found at %s -- build it with `make -C voicefixer_main_amd/csrc`" % TEST_LIB_PATH)
    try:
        # tests only: promote the copy of libvfx.so that load() mapped (whichever path it came from -- VFX_LIB_PATH variant
        # builds included) to the global scope, so that libvfx_test.so binds vfx::plan_resblock, pack_conv, ... to THAT copy
        ctypes.CDLL(LIB_PATH, mode=os.RTLD_NOLOAD | ctypes.RTLD_GLOBAL)
        lib = ctypes.CDLL(TEST_LIB_PATH)
    except OSError as e:
        raise RuntimeError("cannot load %s: %s" % (TEST_LIB_PATH, e))
    for name, (res, args) in TEST_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _test_lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().vfx_last_error()
        raise RuntimeError("%s failed: %s" % (what, msg.decode() if msg else "unknown error"))
