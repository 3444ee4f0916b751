"""ctypes binding of the C ABI declared in include/keraslm_hip.h.

The product path has no CPU fallback: importing this module is harmless (so the
host-side logic can be unit-tested), but `load()` raises if the gfx950 shared
library has not been built, and every compute entry point needs a GPU.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (KL_LIB: another build of the same library, e.g. the -DKL_STAMP diagnostic build or an A/B candidate)
LIB_PATH = os.environ.get("KL_LIB") or os.path.join(os.path.dirname(_HERE), "libkeraslm_hip.so")

KL_PREC_BF16 = 1
KL_PREC_SPLIT = 3


class KlConfig(C.Structure):
    _fields_ = [("depth", C.c_int32), ("width", C.c_int32), ("voc_size", C.c_int32),
                ("n_ctx", C.c_int32), ("ctx_vocab", C.c_int32), ("ctx_dim", C.c_int32)]


class KlError(RuntimeError):
    pass


# name -> (restype, argtypes): every symbol include/keraslm_hip.h declares
SIGNATURES = {
    "kl_abi_version": (C.c_int, []),
    "kl_error_string": (C.c_char_p, [C.c_int]),
    "kl_param_count": (C.c_size_t, [C.POINTER(KlConfig)]),
    "kl_param_layout": (C.c_int, [C.POINTER(KlConfig), C.c_int, C.c_char_p, C.c_size_t,
                                  C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "kl_create": (C.c_void_p, [C.POINTER(KlConfig)]),
    "kl_destroy": (None, [C.c_void_p]),
    "kl_derived_bytes": (C.c_size_t, [C.c_void_p]),
    "kl_bind": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "kl_prepare": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "kl_window_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "kl_forward_window": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "kl_train_window": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "kl_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float,
                               C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "kl_adam_step_scaled": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                     C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "kl_step_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "kl_step_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "kl_host_alloc": (C.c_void_p, [C.c_size_t]),
    "kl_host_free": (None, [C.c_void_p]),
    "kl_step_host_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "kl_step_batch_host": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                     C.c_size_t, C.c_void_p]),
    "kl_step_wait": (C.c_int, [C.c_void_p, C.c_uint32, C.c_double]),
    "kl_state_dist2": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                 C.c_void_p]),
    "kl_trace_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "kl_set_window_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "kl_set_loss_rows": (C.c_int, [C.c_void_p, C.c_int]),
    "kl_trace_kernel_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "kl_trace_read": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_int),
                                C.POINTER(C.c_double)]),
    "kl_test_gemm_tn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                  C.c_long, C.c_long, C.c_long, C.c_int, C.c_int, C.c_void_p]),
    "kl_test_gemm_an": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_long, C.c_long,
                                  C.c_long, C.c_int, C.c_int, C.c_void_p]),
    "kl_test_thin_gemm": (C.c_int, [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int,
                                    C.c_int, C.c_void_p, C.c_long, C.c_int, C.c_void_p]),
}

_lib = None


def load():
    """Load libkeraslm_hip.so (built by `make -C ocrd_keraslm_amd/csrc` or
    __graft_entry__.build()).  Raises if it is missing -- there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KlError("HIP extension %s not built: run `make -C ocrd_keraslm_amd/csrc` "
                      "(the Rater has no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code, what=""):
    if code != 0:
        msg = load().kl_error_string(code).decode()
        raise KlError("%s failed: %s (code %d)" % (what or "keraslm_hip call", msg, code))
