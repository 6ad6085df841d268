"""ctypes binding of libhelmnet_hip.so (C ABI: include/helmnet_hip.h).

There is deliberately NO fallback: if the shared object is missing or cannot be loaded the
import of the product path raises, and every compute entry point of the package goes through
this module.  ``import torch`` must come first so that the HIP runtime already resident in the
process (torch's ``libamdhip64.so.7``) is the one the library binds to -- streams and device
pointers are then shared with PyTorch.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

import torch  # noqa: F401  (must be imported before the library is loaded, see above)

# HELMNET_HIP_LIB: another build of the same library (A/B experiments, tools/build_variant.sh); it must export the same ABI
_LIB_PATH = os.environ.get("HELMNET_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhelmnet_hip.so")
_lib = None

HN_ACT = {"prelu": 0, "relu": 1, "leakyrelu": 2, "celu": 3, "tanh": 4, "gelu": 5, "tanhshrink": 6, "softplus": 7}
# enum hn_precision / hn_option / hn_counter of include/helmnet_hip.h
HN_PRECISION = {"fp32": 0, "bf16x3": 1, "fp16": 2, "bf16x2": 3, "valu": 4}
HN_OPTION = {"lanes": 0, "side_stream": 1, "deep": 3, "spectral_pfa": 4, "spectral_radix16": 5, "dc_valu": 6, "spectral_cols": 7, "train_fused": 10, "train_overlap": 11, "dc_pair": 12, "side_sync": 13, "state_kernel": 14, "hist_copy": 15, "inc_sigma_map": 16,
             "graph": 100, "train_lanes": 101}   # 100 +: laboratory knobs (HN_EXP_*)
HN_COUNTER = {"graph_replays": 0, "eager_iterations": 1, "graphs_captured": 2, "stream_probes": 3, "side_candidate": 4, "train_fwd_events": 5, "flag_sync_iterations": 6}
ABI_VERSION = 7

# name -> (restype, argtypes); every symbol include/helmnet_hip.h declares
SYMBOLS = {
    "hn_abi_version": (c_int, []),
    "hn_create": (c_int, [POINTER(c_void_p), c_int]),
    "hn_destroy": (None, [c_void_p]),
    "hn_last_error": (c_char_p, [c_void_p]),
    "hn_set_unet_precision": (c_int, [c_void_p, c_int]),
    "hn_get_unet_precision": (c_int, [c_void_p]),
    "hn_set_option": (c_int, [c_void_p, c_int, c_int]),
    "hn_get_counter": (c_int64, [c_void_p, c_int]),
    "hn_check_async_errors": (c_int, [c_void_p]),
    "hn_weight_count": (c_size_t, [c_int, c_int, c_int]),
    "hn_load_weights": (c_int, [c_void_p, POINTER(c_float), c_size_t, c_int, c_int, c_int, c_int]),
    "hn_set_domain": (c_int, [c_void_p, c_int, c_int, c_float, c_float]),
    "hn_get_sigmas": (c_int, [c_void_p, c_void_p, c_void_p]),
    "hn_state_len": (c_int64, [c_void_p]),
    "hn_reserve": (c_int, [c_void_p, c_int]),
    "hn_laplacian": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "hn_residual": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "hn_residual_vjp": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "hn_rmse": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "hn_unet": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "hn_double_conv": (c_int, [c_void_p, c_void_p, c_int, c_int, POINTER(c_float), c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "hn_conv8x8": (c_int, [c_void_p, c_void_p, POINTER(c_float), c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "hn_out_conv": (c_int, [c_void_p, c_void_p, POINTER(c_float), c_void_p, c_int, c_int, c_int, c_void_p]),
    "hn_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "hn_train_reserve": (c_int, [c_void_p, c_int, c_int]),
    "hn_train_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float,
                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "hn_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_float, c_float, c_float,
                             c_float, c_float, c_int64, c_void_p]),
    "hn_rows_gather": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p]),
    "hn_rows_scatter": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "hn_train_peek": (c_int64, [c_void_p, c_int, c_int, c_void_p, c_int64, c_void_p]),
    "hn_train_set_forward_event": (c_int, [c_void_p, c_void_p, c_void_p, c_int64]),
    "hn_profile_enable": (c_int, [c_void_p, ctypes.c_uint64]),
    "hn_profile_min": (c_int, [c_void_p, POINTER(ctypes.c_double), c_int]),
    "hn_profile_stride": (c_int, [c_void_p, c_int]),
    "hn_profile_collect": (c_int, [c_void_p, POINTER(ctypes.c_double), POINTER(c_int64), c_int]),
}


class HelmnetHipError(RuntimeError):
    pass


def lib_path() -> str:
    return _LIB_PATH


def load():
    """Load (once) and return the ctypes handle; raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise HelmnetHipError(
            f"{_LIB_PATH} is missing: build it with `python -m helmnet_amd.build` "
            "(hipcc --offload-arch=gfx950).  helmnet_amd has no CPU / PyTorch fallback."
        )
    lib = ctypes.CDLL(_LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export the symbol
        fn.restype = res
        fn.argtypes = args
    if lib.hn_abi_version() != ABI_VERSION:
        raise HelmnetHipError(f"ABI version mismatch: library reports {lib.hn_abi_version()}, binding expects {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, ctx=None, what: str = ""):
    if rc == 0:
        return
    msg = load().hn_last_error(ctx)
    text = msg.decode() if msg else "unknown error"
    if rc == -3:
        raise NotImplementedError(f"{what}: {text}")
    if rc == -1:
        raise ValueError(f"{what}: {text}")
    raise HelmnetHipError(f"{what} failed (status {rc}): {text}")
