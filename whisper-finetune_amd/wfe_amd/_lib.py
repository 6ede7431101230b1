"""ctypes binding of the C ABI in include/wfe.h (libwfe_hip.so, built in-tree by
whisper-finetune_amd/csrc/Makefile).  There is no CPU implementation behind this
module: if the library is missing the import fails, and if no gfx950 device is
present ``wfe_create`` fails -- loudly, both times."""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- loads torch's libamdhip64.so.7 first so that ours binds to the same runtime

_HERE = os.path.dirname(os.path.abspath(__file__))
# WFE_LIB_PATH: a diagnostic build of the same library (stamps, experiments); never a different implementation
LIB_PATH = os.environ.get("WFE_LIB_PATH") or os.path.join(_HERE, "libwfe_hip.so")


class WfeError(RuntimeError):
    pass


class WfeConfig(C.Structure):
    _fields_ = [("n_mel", C.c_int32), ("n_fft", C.c_int32), ("hop_length", C.c_int32),
                ("sampling_rate", C.c_int32), ("device", C.c_int32)]


class WfeAugOp(C.Structure):
    _fields_ = [("kind", C.c_int32), ("i0", C.c_int32), ("i1", C.c_int32), ("i2", C.c_int32),
                ("f0", C.c_float), ("f1", C.c_float), ("d0", C.c_double)]


AUG_SPEED, AUG_SHIFT, AUG_VOLUME, AUG_NOISE, AUG_AFFINE = 1, 2, 3, 4, 5
ERR_ARG, ERR_HIP, ERR_ALLOC, ERR_SAMPLING = -1, -2, -3, -4

# every symbol include/wfe.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "wfe_create": (C.c_int, [C.POINTER(WfeConfig), C.POINTER(_P)]),
    "wfe_destroy": (C.c_int, [_P]),
    "wfe_last_error": (C.c_char_p, [_P]),
    "wfe_mel_filters": (C.c_int, [_P, _P]),
    "wfe_logmel": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, _P, _P]),
    "wfe_logmel_static": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, _P, _P]),
    "wfe_power_spectrum": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, _P, _P]),
    "wfe_augment_logmel": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, _P, _P, C.c_int64,
                                     _P, _P, C.c_int32, _P, _P]),
    "wfe_logmel_pcm16": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_int64,
                                   _P, _P, C.c_int32, _P, _P]),
    "wfe_reverb": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, _P, _P, C.c_int32, _P, _P, C.c_int64, _P]),
    "wfe_resample": (C.c_int, [_P, _P, C.c_int32, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int64,
                               _P, _P]),
    "wfe_resample_geometry": (C.c_int, [C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "wfe_build_resample_table": (C.c_int, [C.c_int32, C.c_int32, _P]),
    "wfe_augment_waveform": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, _P, _P, C.c_int64,
                                       _P, _P, C.c_int32, _P, _P, _P]),
    "wfe_noise_bank_changed": (C.c_int, [_P]),
    "wfe_set_output_format": (C.c_int, [_P, C.c_int32]),
    "wfe_kernel_timing": (C.c_int, [_P, C.c_int32]),
    "wfe_kernel_times": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "wfe_dft_table_floats": (C.c_int64, []),
    "wfe_mel_table_floats": (C.c_int64, []),
    "wfe_build_dft_table": (C.c_int, [_P]),
    "wfe_build_mel_table": (C.c_int, [C.c_int32, _P, _P]),
    "wfe_build_mel_filters": (C.c_int, [C.c_int32, _P]),
    "wfe_dft_n_of": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32]),
    "wfe_version": (C.c_char_p, []),
}

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WfeError(
                f"{LIB_PATH} not found: build it with `make -C whisper-finetune_amd/csrc` "
                "(or __graft_entry__.build()).  There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, handle=None) -> None:
    if rc == 0:
        return
    msg = lib().wfe_last_error(handle)
    text = msg.decode(errors="replace") if msg else ""
    if rc in (ERR_ARG, ERR_SAMPLING):
        raise ValueError(f"wfe: {text}")
    raise WfeError(f"wfe error {rc}: {text}")
