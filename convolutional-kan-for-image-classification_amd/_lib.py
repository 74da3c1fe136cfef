"""ctypes binding of libkanconv.so (C ABI declared in include/kanconv.h).

The library is built in-tree by ``build.py`` (hipcc --offload-arch=gfx950).  There is no
fallback: if the shared object is missing or a call fails, the op raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# KANCONV_LIB: load a measurement variant of the library instead (tools/ab_env.py: a -DKAN_TUNING_KNOBS build next to the shipped one); unset in any normal use
LIB_PATH = os.environ.get("KANCONV_LIB") or os.path.join(_HERE, "libkanconv.so")

KAN_MAX_PLANES = 16
KAN_MAX_TABLE = 32
KAN_FP_WORDS = 192
BASIS_BSPLINE, BASIS_RBF, BASIS_CHEBY, BASIS_POLY, BASIS_FOURIER, BASIS_RELU, BASIS_GRAM = 0, 1, 2, 3, 4, 5, 6
ACT_NONE, ACT_IDENTITY, ACT_GELU, ACT_SILU, ACT_RELU, ACT_TANH, ACT_SIGMOID, ACT_GELU_TANH = -1, 0, 1, 2, 3, 4, 5, 6


class KanGeom(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("B", "C", "H", "W", "O", "Ho", "Wo", "kh", "kw", "sh", "sw", "ph", "pw", "dh", "dw", "groups")] + \
               [("x_bstride", C.c_longlong), ("y_bstride", C.c_longlong)]


class KanBasis(C.Structure):
    _fields_ = [("kind", C.c_int), ("n_basis", C.c_int), ("order", C.c_int), ("act", C.c_int),
                ("p0", C.c_float), ("p1", C.c_float), ("table", C.c_float * KAN_MAX_TABLE), ("chan_table", C.c_void_p)]


class KanPlan(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("P", "K", "IPC", "KC", "Kpad", "Opad", "fwd_splits", "bwd_data_splits", "bwd_weight_splits",
                                          "fwd_target", "bwd_data_target", "bwd_weight_target", "x_pm_wanted", "dz_pm_wanted", "bwd_weight_expanded", "row_blocks", "e_pm_wanted", "fwd_expanded", "fwd_halo", "bwd_weight_halo",
                                          "fwd_band", "bwd_weight_band")] + \
               [(n, C.c_longlong) for n in ("packed_weight_bytes", "bwd_data_weight_bytes", "fwd_slab_elems", "bwd_data_slab_elems",
                                            "bwd_weight_slab_elems", "e_pm_elems")]


class KanWavGeom(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("B", "C", "H", "W", "O", "Ho", "Wo", "kh", "kw", "sh", "sw", "ph", "pw", "dh", "dw", "wavelet")] + \
               [("x_bstride", C.c_longlong), ("u_bstride", C.c_longlong)]


WAVELETS = {"mexican_hat": 0, "morlet": 1, "dog": 2, "meyer": 3, "shannon": 4}

# every symbol include/kanconv.h declares, with its argument types
_P, _I, _LL, _F, _D = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_double
_GP, _BP = C.POINTER(KanGeom), C.POINTER(KanBasis)
SIGNATURES = {
    "kan_version": (C.c_char_p, []),
    "kan_last_error": (C.c_char_p, []),
    "kan_plan": (_I, [_GP, _BP, C.POINTER(KanPlan)]),
    "kan_pack_weights": (_I, [_P, _P, _P, _P, _GP, _BP, _P]),
    "kan_pack_cacheable": (_I, [_GP, _BP]),
    "kan_pack_weights_cached": (_I, [_P, _P, _P, _P, _GP, _BP, _P, _I, _I, _I, _P]),
    "kan_conv_fwd": (_I, [_P, _P, _P, _P, _GP, _BP, _P, _P]),
    "kan_position_major": (_I, [_P, _P, _I, _I, _I, _LL, _P]),
    "kan_conv_bwd_data": (_I, [_P, _P, _P, _P, _P, _P, _GP, _BP, _P, _P]),
    "kan_conv_bwd_data_params": (_I, [_P, _P, _P, _P, _P, _P, _P, _GP, _BP, _P, _P]),
    "kan_conv_bwd_weight": (_I, [_P, _P, _P, _P, _GP, _BP, _P, _P, _P]),
    "kan_position_major_expanded": (_I, [_P, _P, _GP, _BP, _P]),
    "kan_conv_fwd_expanded": (_I, [_P, _P, _P, _GP, _BP, _P]),
    "kan_conv_bwd_weight_expanded": (_I, [_P, _P, _P, _GP, _BP, _P]),
    "kan_unpack_wgrad": (_I, [_P, _P, _P, _GP, _BP, _P]),
    "kan_slab_reduce": (_I, [_P, _I, _LL, _P, _I, _I, _I, _LL, _P]),
    "kan_instnorm_prelu_fwd": (_I, [_P, _I, _LL, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _LL, _F, _I, _P]),
    "kan_instnorm_prelu_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _LL, _I, _P]),
    "kan_instnorm_prelu_pool_fwd": (_I, [_P, _I, _LL, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _LL, _F, _I, _P]),
    "kan_instnorm_prelu_pool_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _LL, _I, _P]),
    "kan_instnorm_prelu_poolk_fwd": (_I, [_P, _I, _LL, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _LL, _F, _I, _I, _I, _P]),
    "kan_instnorm_prelu_poolk_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _LL, _I, _I, _I, _P]),
    "kan_split_supported": (_I, [C.POINTER(KanGeom), C.POINTER(KanBasis)]),
    "kan_split_weight_bytes": (_LL, [C.POINTER(KanGeom), C.POINTER(KanBasis)]),
    "kan_split_pack_weights": (_I, [_P, _P, _P, C.POINTER(KanGeom), C.POINTER(KanBasis), _P]),
    "kan_conv_fwd_split": (_I, [_P, _P, _P, C.POINTER(KanGeom), C.POINTER(KanBasis), _P]),
    "kan_wav_fwd": (_I, [_P, _P, _P, _P, _P, C.POINTER(KanWavGeom), _P]),
    "kan_wav_bwd_input": (_I, [_P, _P, _P, _P, _P, _P, C.POINTER(KanWavGeom), _P]),
    "kan_wav_param_workspace": (_LL, [C.POINTER(KanWavGeom)]),
    "kan_wav_bwd_params": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(KanWavGeom), _P]),
    "kan_adamw_step": (_I, [_P, _P, _P, _P, _LL, _D, _D, _D, _D, _D, _I, _F, _P]),
    "kan_adamw_step_segments": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _D, _D, _D, _D, _D, _I, _F, _P]),
}

_lib: Optional[C.CDLL] = None


class KanConvError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libkanconv.so once; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise KanConvError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU / eager fallback for this path.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().kan_last_error().decode(errors="replace")
        raise KanConvError(f"{what} failed (rc={rc}): {msg}")


def version() -> str:
    return load().kan_version().decode()
