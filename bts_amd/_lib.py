"""ctypes binding of libbts_hip.so (C ABI declared in include/bts_hip.h).

The product path has NO fallback: if the library is missing or a symbol is absent this
module raises at import/first use, and every op raises on non-CUDA tensors.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbts_hip.so")

SYMBOLS = (
    "bts_hip_abi_version", "bts_hip_error_string", "bts_lpg_fwd_f32", "bts_lpg_bwd_f32", "bts_lpg_fused_fwd_f32",
    "bts_reduc_fwd_f32", "bts_conv_fwd_f32", "bts_nchw_to_nhwc_f32", "bts_nhwc_to_nchw_f32",
    "bts_pack_planes_f32", "bts_get_depth_f32", "bts_conv_plan_f32", "bts_conv_plan_ksteps_f32", "bts_maxpool3x3s2_nhwc_f32", "bts_bn_relu_avgpool2_nhwc_f32",
    "bts_conv_wgrad_f32", "bts_bn_train_ws_floats", "bts_bn_train_stats_f32", "bts_bn_apply_nhwc_f32", "bts_bn_train_bwd_f32",
    "bts_pack_weights_blocks", "bts_pack_weights_f32", "bts_pack_wino_floats", "bts_pack_wino_f32", "bts_eval_ws_doubles", "bts_eval_depth_metrics_f32",
    "bts_reduc_lpg_fwd_f32", "bts_plan_run", "bts_upconv_combine_f32",
)

ABI_VERSION = 14


class ConvDesc(C.Structure):
    """struct bts_conv_desc (include/bts_hip.h) -- field order and types must match."""
    _fields_ = [
        ("x", C.c_void_p), ("x_pix_stride", C.c_long), ("c_in_ld", C.c_int), ("k_pad", C.c_int),
        ("B", C.c_int), ("h_in", C.c_int), ("w_in", C.c_int), ("up", C.c_int),
        ("ksize", C.c_int), ("dil", C.c_int), ("stride", C.c_int), ("pad", C.c_int),
        ("w", C.c_void_p), ("c_out", C.c_int), ("c_out_pad", C.c_int),
        ("pre_scale", C.c_void_p), ("pre_shift", C.c_void_p), ("pre_relu", C.c_int),
        ("e1_scale", C.c_void_p), ("e1_shift", C.c_void_p), ("act", C.c_int),
        ("e2_scale", C.c_void_p), ("e2_shift", C.c_void_p),
        ("y", C.c_void_p), ("y_pix_stride", C.c_long), ("y_nchw", C.c_int),
        ("subpixel", C.c_int), ("y2", C.c_void_p), ("y2_pix_stride", C.c_long),
        ("splitk_ws", C.c_void_p), ("splitk_ws_floats", C.c_long),
        ("res", C.c_void_p), ("res_pix_stride", C.c_long), ("n_bundles", C.c_int), ("precision", C.c_int),
        ("tail_planes", C.c_void_p * 4), ("n_tail", C.c_int), ("fill_frames", C.c_int), ("w_split", C.c_void_p), ("w_wino", C.c_void_p),
    ]


class ConvWgradDesc(C.Structure):
    """struct bts_conv_wgrad_desc (include/bts_hip.h)."""
    _fields_ = [
        ("x", C.c_void_p), ("x_pix_stride", C.c_long), ("c_in", C.c_int),
        ("dy", C.c_void_p), ("dy_pix_stride", C.c_long), ("c_out", C.c_int),
        ("B", C.c_int), ("h_in", C.c_int), ("w_in", C.c_int),
        ("up", C.c_int), ("ksize", C.c_int), ("dil", C.c_int), ("stride", C.c_int), ("pad", C.c_int),
        ("dw", C.c_void_p), ("ws", C.c_void_p), ("ws_floats", C.c_long), ("n_bundles", C.c_int),
        ("pre_scale", C.c_void_p), ("pre_shift", C.c_void_p), ("pre_relu", C.c_int),
    ]


class BtsHipError(RuntimeError):
    pass


def source_hash() -> str:
    """sha256 (first 16 hex digits) over the kernel sources the library is built from (csrc/*.hip, common.h,
    include/bts_hip.h): stamps profiles so that a counter value measured on older kernels is never quoted for newer ones."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h"))
                   + glob.glob(os.path.join(_HERE, "csrc", "*.inc")))
    files.append(os.path.join(os.path.dirname(_HERE), "include", "bts_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


_lib = None
_tls = threading.local()


class recording:
    """While active (this thread only), load() hands out `proxy` instead of the library: bts_amd/plan.py records the
    calls a forward makes through it."""

    def __init__(self, proxy):
        self.proxy = proxy

    def __enter__(self):
        self.prev = getattr(_tls, "proxy", None)
        _tls.proxy = self.proxy
        return self.proxy

    def __exit__(self, *exc):
        _tls.proxy = self.prev
        return False


def is_recording() -> bool:
    return getattr(_tls, "proxy", None) is not None


def load():
    """The loaded library (or the recording proxy standing in for it, see `recording`)."""
    proxy = getattr(_tls, "proxy", None)
    return proxy if proxy is not None else load_real()


def load_real():
    """Load the library once; raise loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BtsHipError(
            "bts_amd: %s not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C bts_amd/csrc`; there is no CPU/PyTorch fallback for the hot path" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for s in SYMBOLS:
        if not hasattr(lib, s):
            raise BtsHipError("bts_amd: symbol %s missing from %s" % (s, LIB_PATH))
    vp, i, l, f = C.c_void_p, C.c_int, C.c_long, C.c_float
    lib.bts_hip_abi_version.restype = i
    lib.bts_hip_abi_version.argtypes = []
    lib.bts_hip_error_string.restype = C.c_char_p
    lib.bts_hip_error_string.argtypes = [i]
    lib.bts_lpg_fwd_f32.restype = i
    lib.bts_lpg_fwd_f32.argtypes = [vp, i, i, i, i, vp, vp, vp]
    lib.bts_lpg_bwd_f32.restype = i
    lib.bts_lpg_bwd_f32.argtypes = [vp, vp, i, i, i, i, vp, vp]
    lib.bts_lpg_fused_fwd_f32.restype = i
    lib.bts_lpg_fused_fwd_f32.argtypes = [vp, i, i, i, i, i, f, vp, vp, i, l, vp, vp]
    lib.bts_reduc_fwd_f32.restype = i
    lib.bts_reduc_fwd_f32.argtypes = [vp, l, l, i, i, vp, l, f, i, i, vp, vp]
    lib.bts_reduc_lpg_fwd_f32.restype = i
    lib.bts_reduc_lpg_fwd_f32.argtypes = [vp, l, i, i, i, i, i, vp, l, f, i, vp, vp, vp, vp, vp]
    lib.bts_conv_fwd_f32.restype = i
    lib.bts_conv_fwd_f32.argtypes = [C.POINTER(ConvDesc), vp]
    lib.bts_conv_wgrad_f32.restype = i
    lib.bts_conv_wgrad_f32.argtypes = [C.POINTER(ConvWgradDesc), vp]
    lib.bts_bn_train_ws_floats.restype = l
    lib.bts_bn_train_ws_floats.argtypes = [l, i]
    lib.bts_bn_train_stats_f32.restype = i
    lib.bts_bn_train_stats_f32.argtypes = [vp, l, l, i, vp, vp, f, f, vp, vp, vp, l, vp, vp, vp, vp, vp]
    lib.bts_bn_apply_nhwc_f32.restype = i
    lib.bts_bn_apply_nhwc_f32.argtypes = [vp, l, l, i, vp, vp, i, vp, l, vp]
    lib.bts_bn_train_bwd_f32.restype = i
    lib.bts_bn_train_bwd_f32.argtypes = [vp, l, vp, l, l, i, vp, vp, vp, vp, i, vp, l, vp, vp, vp, l, vp]
    lib.bts_pack_weights_blocks.restype = l
    lib.bts_pack_weights_blocks.argtypes = [l, l]
    lib.bts_pack_weights_f32.restype = i
    lib.bts_pack_weights_f32.argtypes = [vp, i, l, vp]
    lib.bts_pack_wino_floats.restype = l
    lib.bts_pack_wino_floats.argtypes = [i, i, i, i]
    lib.bts_pack_wino_f32.restype = i
    lib.bts_pack_wino_f32.argtypes = [vp, i, l, i, i, i, vp, vp]
    lib.bts_conv_plan_f32.restype = i
    lib.bts_conv_plan_f32.argtypes = [C.POINTER(ConvDesc), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.bts_conv_plan_ksteps_f32.restype = i
    lib.bts_conv_plan_ksteps_f32.argtypes = [C.POINTER(ConvDesc), C.POINTER(C.c_long), C.POINTER(C.c_long)]
    lib.bts_nchw_to_nhwc_f32.restype = i
    lib.bts_nchw_to_nhwc_f32.argtypes = [vp, i, i, l, vp, l, i, vp]
    lib.bts_nhwc_to_nchw_f32.restype = i
    lib.bts_nhwc_to_nchw_f32.argtypes = [vp, l, i, i, l, vp, vp]
    lib.bts_maxpool3x3s2_nhwc_f32.restype = i
    lib.bts_maxpool3x3s2_nhwc_f32.argtypes = [vp, l, i, i, i, i, vp, l, vp, l, vp]
    lib.bts_bn_relu_avgpool2_nhwc_f32.restype = i
    lib.bts_bn_relu_avgpool2_nhwc_f32.argtypes = [vp, l, i, i, i, i, vp, vp, vp, l, vp]
    lib.bts_pack_planes_f32.restype = i
    lib.bts_pack_planes_f32.argtypes = [vp, vp, vp, vp, i, l, vp, l, vp]
    lib.bts_get_depth_f32.restype = i
    lib.bts_get_depth_f32.argtypes = [vp, vp, i, i, i, i, f, vp, vp, vp]
    lib.bts_upconv_combine_f32.restype = i
    lib.bts_upconv_combine_f32.argtypes = [vp, l, i, i, i, i, vp, vp, i, vp, l, vp]
    lib.bts_eval_ws_doubles.restype = l
    lib.bts_eval_ws_doubles.argtypes = [i, i, i]
    lib.bts_eval_depth_metrics_f32.restype = i
    lib.bts_eval_depth_metrics_f32.argtypes = [vp, i, i, i, vp, i, i, i, i, f, f, i, i, i, i, vp, l, vp, vp, vp]
    if lib.bts_hip_abi_version() != ABI_VERSION:
        raise BtsHipError("bts_amd: ABI version mismatch (%d != %d)" % (lib.bts_hip_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


TORCH_LIB_PATH = os.path.join(_HERE, "libbts_torch.so")
_torch_ops = None


def load_torch_ops():
    """``torch.ops.bts_hip``: the TORCH_LIBRARY operator shell over the C ABI (csrc/torch_ops.cpp -> libbts_torch.so:
    TORCH_CHECK validation, device guard, current stream).  Loaded once; raises loudly if it is not built."""
    global _torch_ops
    if _torch_ops is not None:
        return _torch_ops
    import torch
    load_real()                                    # the C ABI library first: libbts_torch.so links against it
    if not os.path.exists(TORCH_LIB_PATH):
        raise BtsHipError("bts_amd: %s not found -- build it with `make -C bts_amd/csrc` (or BTS_BINDING=ctypes to bind the "
                          "C ABI directly)" % TORCH_LIB_PATH)
    torch.ops.load_library(TORCH_LIB_PATH)
    _torch_ops = torch.ops.bts_hip
    return _torch_ops


def check(code: int, what: str):
    if code != 0:
        msg = load().bts_hip_error_string(code)
        raise BtsHipError("%s failed: %s (code %d)" % (what, msg.decode() if msg else "?", code))
