"""ctypes binding of libtcvn_hip.so (the C ABI declared in include/tcvn_hip.h).

The library is built in-tree by ``make -C dune-transformercvn_amd/csrc`` (or ``__graft_entry__.build()``).  There is no
CPU fallback: if the shared object is missing or fails to load, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- first: the library must bind to the HIP runtime PyTorch ships (one runtime per process); loading
#                              /opt/rocm's copy ahead of torch's leaves this library without a device ("no ROCm-capable device")

from . import _libselect

_HERE = os.path.dirname(os.path.abspath(__file__))
# libtcvn_hip.so unless a test child process selected the debug build explicitly (_libselect.use) before this import; no
# environment variable is read here or in the product library.
LIB_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "lib", _libselect.NAME))
_libselect._bound = True

MODE_F32, MODE_BF16 = 0, 1
SLOT_PARAM, SLOT_BUFFER, SLOT_COUNTER = 0, 1, 2


class DenseNetCfg(C.Structure):
    _fields_ = [("in_ch", C.c_int), ("out_dim", C.c_int), ("init_ch", C.c_int), ("growth", C.c_int), ("bn_size", C.c_int),
                ("n_blocks", C.c_int), ("layers", C.c_int * 8), ("H", C.c_int), ("W", C.c_int), ("dropout", C.c_float),
                ("mode", C.c_int)]


class SdxlCfg(C.Structure):
    _fields_ = [("in_ch", C.c_int), ("out_dim", C.c_int), ("init_ch", C.c_int), ("repeat", C.c_int), ("num_blocks", C.c_int),
                ("H", C.c_int), ("W", C.c_int), ("mode", C.c_int)]


class HeadCfg(C.Structure):
    _fields_ = [("hidden_dim", C.c_int), ("heads", C.c_int), ("n_layers", C.c_int), ("in_dim", C.c_int),
                ("event_classes", C.c_int), ("prong_classes", C.c_int), ("n_dec", C.c_int), ("dec_dims", C.c_int * 8),
                ("dec_out_in", C.c_int), ("gelu", C.c_int), ("norm_first", C.c_int), ("dropout_modules", C.c_int),
                ("dropout", C.c_float), ("gamma", C.c_float), ("event_weight", C.c_float),
                ("no_linear_bn", C.c_int), ("linear_relu", C.c_int)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"libtcvn_hip.so not found at {LIB_PATH}: build it with `make -C dune-transformercvn_amd/csrc -j8` "
            "(hipcc, --offload-arch=gfx950). There is no CPU fallback for the TransformerCVN hot path.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, u64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float
    P = C.POINTER

    def sig(name, res, *args):
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = list(args)
        return fn

    sig("tcvn_version", i32)
    sig("tcvn_densenet_create", i32, P(DenseNetCfg), P(vp))
    sig("tcvn_densenet_destroy", None, vp)
    sig("tcvn_densenet_num_slots", i32, vp)
    sig("tcvn_densenet_slot", i32, vp, i32, C.c_char_p, i32, P(i64), P(i32))
    sig("tcvn_densenet_bind", i32, vp, P(vp), P(vp))
    sig("tcvn_densenet_workspace_bytes", i64, vp, i32, i32)
    sig("tcvn_densenet_forward", i32, vp, i32, vp, vp, i64, i32, f32, vp, i64, vp, i64, i32, u64, vp)
    sig("tcvn_densenet_backward", i32, vp, i32, vp, i64, vp, i64, vp)
    sig("tcvn_densenet_num_blocks", i32, vp)
    sig("tcvn_densenet_backward_blocks", i32, vp, i32, vp, i64, vp, i64, i32, i32, vp)
    sig("tcvn_densenet_tap", i32, vp, i32, C.c_char_p, P(i64), P(i32), P(i32), P(i32), P(i32), P(i32), P(i32))
    sig("tcvn_sdxl_create", i32, P(SdxlCfg), P(vp))
    sig("tcvn_sdxl_destroy", None, vp)
    sig("tcvn_sdxl_num_slots", i32, vp)
    sig("tcvn_sdxl_slot", i32, vp, i32, C.c_char_p, i32, P(i64), P(i32))
    sig("tcvn_sdxl_bind", i32, vp, P(vp), P(vp))
    sig("tcvn_sdxl_workspace_bytes", i64, vp, i32, i32)
    sig("tcvn_sdxl_forward", i32, vp, i32, vp, vp, i64, i32, f32, vp, i64, vp, i64, i32, u64, vp)
    sig("tcvn_sdxl_backward", i32, vp, i32, vp, i64, vp, i64, vp)
    sig("tcvn_sdxl_tap", i32, vp, i32, C.c_char_p, P(i64), P(i32), P(i32), P(i32), P(i32), P(i32), P(i32))
    sig("tcvn_head_create", i32, P(HeadCfg), P(vp))
    sig("tcvn_head_destroy", None, vp)
    sig("tcvn_head_num_slots", i32, vp)
    sig("tcvn_head_slot", i32, vp, i32, C.c_char_p, i32, P(i64), P(i32))
    sig("tcvn_head_bind", i32, vp, P(vp), P(vp))
    sig("tcvn_head_workspace_bytes", i64, vp, i32, i32, i32)
    sig("tcvn_head_forward", i32, vp, i32, i32, i32, vp, vp, vp, vp, vp, i64, i32, u64, vp)
    sig("tcvn_head_loss", i32, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp)
    sig("tcvn_head_backward", i32, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, i64, vp)
    sig("tcvn_head_set_fused_encoder", None, vp, i32)
    sig("tcvn_head_embed", i32, vp, i32, i32, i32, vp, vp, vp, vp, i64, i32, u64, vp)
    sig("tcvn_head_encode", i32, vp, i32, i32, vp, vp, vp, vp, i64, i32, u64, vp)
    sig("tcvn_head_decode", i32, vp, i32, i32, vp, vp, vp, vp, i64, i32, u64, vp)
    sig("tcvn_linear_forward", i32, vp, i64, vp, vp, vp, i64, i32, i32, i32, vp)
    sig("tcvn_rows_bn_prelu_forward", i32, vp, i64, i32, i32, vp, vp, vp, vp, vp, vp, i64, vp, i32, f32, u64, C.c_uint32, vp)
    sig("tcvn_linear_backward", i32, vp, i64, vp, i64, vp, vp, i64, vp, vp, i32, i32, i32, vp)
    sig("tcvn_rows_bn_prelu_backward", i32, vp, i64, vp, i64, i32, i32, vp, vp, vp, vp, vp, i64, vp, vp, vp, f32, u64, C.c_uint32, vp)
    sig("tcvn_focal_loss", i32, vp, vp, i32, i32, f32, f32, vp, vp, vp)
    sig("tcvn_grad_sumsq", i32, vp, i64, vp, i32, vp, vp)
    sig("tcvn_adamw_step", i32, vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, i64, vp, f32, vp)
    sig("tcvn_dropout_keep", i32, i32, f32, u64, C.c_uint32, i64, i32, vp, vp)
    sig("tcvn_backward_overlap", None, i32)
    sig("tcvn_profile_enable", None, i32)
    sig("tcvn_profile_filter", None, C.c_char_p)
    sig("tcvn_profile_reset", None)
    sig("tcvn_profile_count", i32)
    sig("tcvn_profile_get", i32, i32, C.c_char_p, i32, P(f32), P(C.c_double), P(C.c_double))
    return lib


def profile_records():
    """[(kernel name, ms, flops, bytes)] of the launches recorded since the last reset (blocks until they finished)."""
    out = []
    buf = C.create_string_buffer(128)
    ms, fl, by = C.c_float(), C.c_double(), C.c_double()
    for i in range(lib.tcvn_profile_count()):
        check(lib.tcvn_profile_get(i, buf, 128, C.byref(ms), C.byref(fl), C.byref(by)), "profile_get")
        out.append((buf.value.decode(), ms.value, fl.value, by.value))
    return out


EXPORTS = [
    "tcvn_grad_sumsq", "tcvn_adamw_step", "tcvn_backward_overlap", "tcvn_profile_enable", "tcvn_profile_filter", "tcvn_profile_reset", "tcvn_profile_count", "tcvn_profile_get",
    "tcvn_focal_loss", "tcvn_dropout_keep", "tcvn_head_set_fused_encoder", "tcvn_head_embed", "tcvn_head_encode", "tcvn_head_decode", "tcvn_linear_forward",
    "tcvn_rows_bn_prelu_forward", "tcvn_linear_backward", "tcvn_rows_bn_prelu_backward", "tcvn_sdxl_create", "tcvn_sdxl_destroy", "tcvn_sdxl_num_slots", "tcvn_sdxl_slot", "tcvn_sdxl_bind",
    "tcvn_sdxl_workspace_bytes", "tcvn_sdxl_forward", "tcvn_sdxl_backward", "tcvn_sdxl_tap",
    "tcvn_version", "tcvn_densenet_create", "tcvn_densenet_destroy", "tcvn_densenet_num_slots", "tcvn_densenet_slot",
    "tcvn_densenet_bind", "tcvn_densenet_workspace_bytes", "tcvn_densenet_forward", "tcvn_densenet_backward",
    "tcvn_densenet_tap", "tcvn_densenet_num_blocks", "tcvn_densenet_backward_blocks", "tcvn_head_create", "tcvn_head_destroy", "tcvn_head_num_slots", "tcvn_head_slot", "tcvn_head_bind",
    "tcvn_head_workspace_bytes", "tcvn_head_forward", "tcvn_head_loss", "tcvn_head_backward",
]

lib = _load()


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"libtcvn_hip: {what} failed with code {rc}")
