"""ctypes binding of libinr_mi355x.so -- the C-ABI declared in include/inr_abi.h.

There is NO fallback: if the shared library is missing or a call fails, a RuntimeError is raised
with the library's own message (inr_last_error).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("INR_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "lib", "libinr_mi355x.so")

# enums (include/inr_abi.h)
KIND_SIREN, KIND_FFN, KIND_WIRE, KIND_FOURIER, KIND_MSFOURIER, KIND_MSBOUNDED, KIND_GABOR, KIND_KGABOR, KIND_WIRE2D = range(9)
PRECISION_F32, PRECISION_BF16 = 0, 1
ACT_ID, ACT_SIN, ACT_TANH, ACT_RELU, ACT_SIGMOID = 0, 1, 2, 3, 4
ACT_CTANH = 7  # WIRE2D last_tanh: complex Tanh before .real
INPUT_X, INPUT_GAUSS = 0, 1
LOSS_L2_HALF, LOSS_L1_HALF, LOSS_TANH, LOSS_LOGSPACE, LOSS_HDR, LOSS_MSLE_HALF, LOSS_CENTER = 0, 1, 2, 3, 4, 5, 6
LOSS_WORDS = 512  # INR_LOSS_WORDS: floats a loss_out buffer must hold (word 0 = loss, the rest ordered partial sums)


class NetDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("in_features", C.c_int32), ("width", C.c_int32), ("depth", C.c_int32),
                ("out_features", C.c_int32), ("last_act", C.c_int32), ("input", C.c_int32),
                ("enc_size", C.c_int32), ("w0", C.c_float), ("first_omega_0", C.c_float),
                ("hidden_omega_0", C.c_float), ("scale_0", C.c_float), ("precision", C.c_int32),
                ("reserved", C.c_int32 * 3)]


class LossDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("eps", C.c_float), ("sigma", C.c_float), ("factor", C.c_float),
                ("inv_count", C.c_float), ("hdr_A", C.c_float), ("scale", C.c_float), ("cons_w", C.c_float),
                ("cons_chan", C.c_int32), ("cons_lo", C.c_float * 4), ("cons_hi", C.c_float * 4),
                ("cons_inv", C.c_float * 4)]


ABI_VERSION = 7  # INR_ABI_VERSION of include/inr_abi.h


class Workspace(C.Structure):
    """inr_workspace: a call's caller-owned scratch with its extents (floats)."""
    _fields_ = [("save", C.c_void_p), ("save_floats", C.c_int64), ("slabs", C.c_void_p), ("slab_floats", C.c_int64)]


class Sizes(C.Structure):
    _fields_ = [("n_params", C.c_int64), ("packed_floats", C.c_int64), ("tile_rows", C.c_int64),
                ("save_bytes_per_tile", C.c_int64), ("max_blocks", C.c_int64), ("slab_floats", C.c_int64),
                ("step_save_by_tile", C.c_int64)]


class StepInfo(C.Structure):
    """inr_step_info: which kernel runs a batch's fused step, and how its tiles are dealt"""
    _fields_ = [("row_split", C.c_int32), ("ncb", C.c_int32), ("grid", C.c_int32), ("rounds", C.c_int32),
                ("hi", C.c_int32), ("lo", C.c_int32), ("n_hi", C.c_int32), ("reserved", C.c_int32)]


# every symbol include/inr_abi.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "inr_abi_version": (C.c_int, []),
    "inr_last_error": (C.c_int, [C.c_char_p, C.c_size_t]),
    "inr_plan_create": (C.c_int, [C.POINTER(NetDesc), C.POINTER(_P)]),
    "inr_plan_destroy": (C.c_int, [_P]),
    "inr_plan_sizes": (C.c_int, [_P, C.POINTER(Sizes)]),
    "inr_plan_launch_dims": (C.c_int, [_P, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "inr_plan_workspace": (C.c_int, [_P, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "inr_plan_step_info": (C.c_int, [_P, C.c_int64, C.POINTER(StepInfo)]),
    "inr_plan_grad_scale_state": (C.c_int, [_P, C.POINTER(C.c_float), _P]),
    "inr_pack_params": (C.c_int, [_P, _P, _P, _P]),
    "inr_encode_gauss": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, _P]),
    "inr_encode_logf": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, _P]),
    "inr_forward": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, _P, C.POINTER(Workspace), _P]),
    "inr_backward": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, _P, C.POINTER(Workspace), _P, _P]),
    "inr_loss_grad": (C.c_int, [C.POINTER(LossDesc), _P, _P, _P, _P, C.c_int64, _P, _P, _P]),
    "inr_loss_grad_multi": (C.c_int, [C.POINTER(LossDesc), _P, _P, _P, _P, C.c_int32, C.c_int64, _P, _P, _P]),
    "inr_loss_tv_grad": (C.c_int, [C.POINTER(LossDesc), _P, _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_float,
                                   _P, _P, _P]),
    "inr_tv_grad": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_float, _P, _P, _P]),
    "inr_center_pairs_grad": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int64, C.c_float, _P, _P, _P]),
    "inr_train_step": (C.c_int, [_P, C.POINTER(LossDesc), _P, _P, _P, _P, _P, _P, C.c_int64, C.POINTER(Workspace), _P,
                                 _P, _P]),
    "inr_plan_heads": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "inr_plan_set_bounds": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int32]),
    "inr_forward_multi": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, _P, C.POINTER(Workspace), C.c_int32, _P]),
    "inr_backward_multi": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, _P, C.POINTER(Workspace), _P, _P]),
    "inr_train_step_multi": (C.c_int, [_P, C.POINTER(LossDesc), _P, _P, _P, _P, _P, _P, _P, C.c_int64,
                                       C.POINTER(Workspace), _P, _P, _P]),
    "inr_adam_step": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_double, C.c_double, C.c_double, C.c_double,
                                C.c_double, C.c_double, C.c_double, C.c_int32, _P]),
    "inr_reg_grad": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, C.c_double, C.c_double, _P, _P]),
    "inr_adam_step_shard": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_double,
                                      C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, _P]),
    "inr_train_adam_step": (C.c_int, [_P, C.POINTER(LossDesc), _P, _P, _P, _P, _P, _P, C.c_int64, C.POINTER(Workspace),
                                      _P, _P, _P, _P, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                      C.c_double, C.c_double, C.c_int32, _P]),
    "inr_adam_step_dev": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int32, _P, C.c_double, C.c_double, C.c_double,
                                    C.c_double, C.c_double, C.c_double, _P]),
    "inr_adam_schedule": (C.c_int, [C.c_double, C.c_double, C.c_double, C.c_int32, _P]),
}

_lib = None


def load() -> C.CDLL:
    """Load the engine.  Raises loudly when the in-tree library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C mri-implicit-neural-representations_amd/csrc`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    if lib.inr_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libinr_mi355x.so ABI version {lib.inr_abi_version()} != {ABI_VERSION}")
    _lib = lib
    return lib


def last_error() -> str:
    buf = C.create_string_buffer(512)
    load().inr_last_error(buf, 512)
    return buf.value.decode()


def check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError(f"libinr_mi355x: error {rc}: {last_error()}")
