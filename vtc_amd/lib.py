"""ctypes binding of libvtc_amd.so (include/vtc_amd.h).

There is no CPU fallback: if the shared library is missing or cannot be loaded this module raises,
and every product entry point that needs it fails loudly.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

ABI_VERSION = 1
MAX_LABELS = 8
N_LAYERS = 12
PROF_CLASSES = 8
STATIC_DECODE_MAX_FRAMES = 262144

OK, ERR_ARG, ERR_HIP, ERR_NO_GPU, ERR_CAPACITY, ERR_SHORT, ERR_NONFINITE = 0, -1, -2, -3, -4, -5, -6

_f32p = C.POINTER(C.c_float)


class VtcSpec(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("n_labels", C.c_int32),
        ("window_samples", C.c_int32),
        ("max_batch", C.c_int32),
        ("mfma_dtype", C.c_int32),
        ("reserved", C.c_int32 * 3),
    ]


_LAYER_FIELDS = ["q_w", "q_b", "k_w", "k_b", "v_w", "v_b", "o_w", "o_b", "ln1_w", "ln1_b",
                 "ff1_w", "ff1_b", "ff2_w", "ff2_b", "ln2_w", "ln2_b"]


class VtcLayerWeights(C.Structure):
    _fields_ = [(n, _f32p) for n in _LAYER_FIELDS]


class VtcWeights(C.Structure):
    _fields_ = [
        ("conv_w", _f32p * 7),
        ("gn_w", _f32p), ("gn_b", _f32p),
        ("fp_ln_w", _f32p), ("fp_ln_b", _f32p),
        ("fp_w", _f32p), ("fp_b", _f32p),
        ("pos_w", _f32p), ("pos_b", _f32p),
        ("enc_ln_w", _f32p), ("enc_ln_b", _f32p),
        ("layer", VtcLayerWeights * N_LAYERS),
        ("head_w", _f32p), ("head_b", _f32p),
    ]


# every symbol include/vtc_amd.h declares: name -> (restype, argtypes)
_i64p, _i32p, _f64p = C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_double)
SYMBOLS = {
    "vtc_create": (C.c_int, [C.POINTER(VtcSpec), C.POINTER(VtcWeights), C.c_int, C.POINTER(C.c_void_p)]),
    "vtc_destroy": (None, [C.c_void_p]),
    "vtc_last_error": (C.c_char_p, [C.c_void_p]),
    "vtc_frames_per_window": (C.c_int, [C.c_void_p]),
    "vtc_frames_for_samples": (C.c_int64, [C.c_int64]),
    "vtc_forward_windows": (C.c_int, [C.c_void_p, C.c_void_p, _i64p, _i32p, _i64p, _i32p, C.c_int32, C.c_void_p,
                                      C.c_void_p]),
    "vtc_infer_waveform": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, _i64p, C.c_void_p]),
    "vtc_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, _f32p, _f32p, _i64p, _i64p, _i32p, C.c_int64, _i64p,
                             _f32p, _f64p, C.c_void_p]),
    "vtc_decode_static": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, _f32p, _f32p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "vtc_threshold_mask": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, _f32p, _f32p, C.c_void_p, C.c_void_p]),
    "vtc_infer_waveforms": (C.c_int, [C.c_void_p, C.c_void_p, _i64p, _i64p, C.c_int32, C.c_int32, C.c_void_p, _i64p, C.c_void_p]),
    "vtc_decode_files": (C.c_int, [C.c_void_p, C.c_void_p, _i64p, C.c_int32, _f32p, _f32p, _i32p, _i64p, _i64p, _i32p, C.c_int64,
                                   _i64p, _f32p, _f64p, C.c_void_p]),
    "vtc_decode_launch": (C.c_int, [C.c_void_p, C.c_void_p, _i64p, C.c_int32, _f32p, _f32p, C.c_int32, C.c_void_p]),
    "vtc_decode_collect": (C.c_int, [C.c_void_p, _i32p, _i64p, _i64p, _i32p, C.c_int64, _i64p, _f32p, _f64p]),
    "vtc_resampled_length": (C.c_int64, [C.c_int64, C.c_int32]),
    "vtc_prepare_audio": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                    _i64p, C.c_void_p]),
    "vtc_flac_info": (C.c_int, [C.c_void_p, C.c_int64, _i32p, _i32p, _i32p, _i64p, C.c_void_p]),
    "vtc_flac_decode": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int64, _i64p]),
    "vtc_flac_decode_threads": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int64, _i64p, C.c_int32]),
    "vtc_ima_adpcm_decode": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, _i64p]),
    "vtc_debug_stage": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32, _f32p, C.c_int64]),
    "vtc_debug_stop_after": (C.c_int, [C.c_void_p, C.c_int32]),
    "vtc_profile_enable": (C.c_int, [C.c_void_p, C.c_int32]),
    "vtc_profile_read": (C.c_int, [C.c_void_p, _f64p, _i64p, _f64p, _f64p]),
    "vtc_profile_class_name": (C.c_char_p, [C.c_int32]),
}

LIB_PATH = Path(os.environ.get("VTC_AMD_LIB", Path(__file__).resolve().parent / "libvtc_amd.so"))
_lib = None


class VtcError(RuntimeError):
    """Non-zero status from libvtc_amd.so.  A plain Exception subclass so the host's per-file
    `except Exception` (reference src/pipeline/vtc.py:245-249) turns it into an error row."""

    def __init__(self, status: int, message: str):
        super().__init__(message)
        self.status = status


def load() -> C.CDLL:
    """Load the HIP library once; raise if it is missing (never fall back to a CPU path)."""
    global _lib
    if _lib is None:
        # torch first: PyTorch-ROCm ships its own libamdhip64; libvtc_amd.so must bind to that one so that
        # the process holds a single HIP/HSA runtime (two runtimes in one process see no devices).
        import torch  # noqa: F401
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} not found: build it with `make -C vtc_amd/csrc` (or __graft_entry__.build()); "
                "vtc_amd has no CPU fallback")
        lib = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def last_error(handle) -> str:
    msg = load().vtc_last_error(handle)
    return msg.decode("utf-8", "replace") if msg else ""


def check(status: int, handle=None) -> None:
    if status != OK:
        raise VtcError(status, last_error(handle) or f"libvtc_amd status {status}")
