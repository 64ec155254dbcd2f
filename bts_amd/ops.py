"""Host-side wrappers over the C ABI (include/bts_hip.h): argument validation, weight packing,
stream plumbing.  torch is used only for device memory and the current stream.

Every function requires CUDA(ROCm) fp32 tensors and raises otherwise -- no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import BtsHipError, ConvDesc, ConvWgradDesc

ACT_NONE, ACT_RELU, ACT_ELU, ACT_SIGMOID = 0, 1, 2, 3


class KernelTrace:
    """Optional per-launch timing with HIP events on the launch stream (bench.py's roofline leg).
    Each record: (kernel, tag, algorithmic flops, algorithmic bytes, start event, end event, executed flops).
    Algorithmic = the reference formulation's op count; executed = what the kernel's own formulation issues
    (they differ for the sub-pixel upconv, which needs 4 taps where the reference spends 9)."""

    def __init__(self):
        self.records = []

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for kern, tag, flops, nbytes, s, e, xflops in self.records:
            d = out.setdefault(kern, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0, xflops=0.0, tags={}))
            ms = s.elapsed_time(e)
            d["launches"] += 1
            d["ms"] += ms
            d["flops"] += flops
            d["bytes"] += nbytes
            d["xflops"] += xflops
            t = d["tags"].setdefault(tag, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0, xflops=0.0))
            t["launches"] += 1
            t["ms"] += ms
            t["flops"] += flops
            t["bytes"] += nbytes
            t["xflops"] += xflops
        return out


_trace: Optional[KernelTrace] = None


def set_trace(t: Optional[KernelTrace]):
    global _trace
    _trace = t


def _launch(kern: str, tag: str, flops: float, nbytes: float, fn, xflops: Optional[float] = None):
    if _trace is None:
        return fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    rc = fn()
    e.record()
    _trace.records.append((kern, tag, flops, nbytes, s, e, flops if xflops is None else xflops))
    return rc


def _stream(t: torch.Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _need(t: torch.Tensor, name: str):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise BtsHipError("bts_amd.%s: expected a CUDA/ROCm tensor (the hot path has no CPU fallback)" % name)
    if t.dtype != torch.float32:
        raise BtsHipError("bts_amd.%s: expected float32, got %s" % (name, t.dtype))


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _rows2d(t: torch.Tensor, name: str) -> Tuple[int, int]:
    """A [npix, C] NHWC view: unit channel stride, arbitrary (>= C) pixel stride."""
    _need(t, name)
    if t.dim() != 2 or t.stride(1) != 1 or t.stride(0) < t.shape[1]:
        raise BtsHipError("bts_amd.%s: expected a [npix, C] view with unit channel stride" % name)
    return t.stride(0), t.shape[1]


# ------------------------------------------------------------------------------ binding of the hot-path operators
# The hot-path operators (LPG, reduction_1x1, reduction -> LPG, the fused convolution) are TORCH OPERATORS:
# torch.ops.bts_hip.* (csrc/torch_ops.cpp, a TORCH_LIBRARY shell over the C ABI -- what the reference's native side does
# with REGISTER_OP / OpKernel, local_planar_guidance.cc:31-72, 116-156, 234-239).  BTS_BINDING=ctypes binds the C ABI
# directly instead (A/B, and what a plan recording listens on: bts_amd/plan.py records through the ctypes proxy).
_BINDING = os.environ.get("BTS_BINDING", "torch").strip().lower()
_autograd_registered = False


def torch_ops():
    """torch.ops.bts_hip, or None when this call must go through the ctypes binding."""
    global _autograd_registered
    if _BINDING != "torch" or _lib.is_recording():
        return None
    t = _lib.load_torch_ops()
    if not _autograd_registered:
        _autograd_registered = True

        def _setup(ctx, inputs, output):
            ctx.save_for_backward(inputs[0])
            ctx.upratio = int(inputs[1])

        def _backward(ctx, grad_depth, grad_abs_min):
            return t.lpg_backward(ctx.saved_tensors[0], grad_depth.contiguous(), ctx.upratio), None

        torch.library.register_autograd("bts_hip::lpg", _backward, setup_context=_setup)
    return t


def _op(fn):
    """Run a torch operator; its TORCH_CHECK failures surface as BtsHipError like the ctypes binding's return codes."""
    try:
        fn()
    except BtsHipError:
        raise
    except RuntimeError as e:
        raise BtsHipError(str(e).split("\n")[0]) from None
    return 0


# ------------------------------------------------------------------------------ LPG
def lpg_forward(plane_eq: torch.Tensor, upratio: int, abs_min: Optional[torch.Tensor] = None) -> torch.Tensor:
    """local_planar_guidance.forward (reference bts.py:149-173): [B,4,h,w] -> [B,h*k,w*k]."""
    _need(plane_eq, "lpg_forward")
    if plane_eq.dim() != 4 or plane_eq.shape[1] != 4:
        raise BtsHipError("lpg_forward: plane_eq must be [B,4,h,w]")
    tops = torch_ops()
    if tops is not None:                         # differentiable (autograd registered on bts_hip::lpg)
        box = []
        _op(lambda: box.append(tops.lpg(plane_eq, int(upratio))))
        out, am = box[0]
        if abs_min is not None:
            abs_min.copy_(am.detach())
        return out
    plane_eq = plane_eq.contiguous()
    B, _, h, w = plane_eq.shape
    k = int(upratio)
    out = torch.empty((B, h * k, w * k), dtype=torch.float32, device=plane_eq.device)
    with torch.cuda.device(plane_eq.device):
        rc = _lib.load().bts_lpg_fwd_f32(_ptr(plane_eq), B, h, w, k, _ptr(out), _ptr(abs_min), _stream(plane_eq))
    _lib.check(rc, "bts_lpg_fwd_f32")
    return out


def lpg_backward(plane_eq: torch.Tensor, grad_depth: torch.Tensor, upratio: int) -> torch.Tensor:
    """Gradient of local_planar_guidance.forward w.r.t. plane_eq (what autograd through bts.py:149-173 yields)."""
    _need(plane_eq, "lpg_backward")
    _need(grad_depth, "lpg_backward")
    plane_eq = plane_eq.contiguous()
    grad_depth = grad_depth.contiguous()
    B, _, h, w = plane_eq.shape
    k = int(upratio)
    if tuple(grad_depth.shape) != (B, h * k, w * k):
        raise BtsHipError("lpg_backward: grad_depth must be [B,h*k,w*k]")
    g = torch.empty_like(plane_eq)
    with torch.cuda.device(plane_eq.device):
        rc = _lib.load().bts_lpg_bwd_f32(_ptr(plane_eq), _ptr(grad_depth), B, h, w, k, _ptr(g), _stream(plane_eq))
    _lib.check(rc, "bts_lpg_bwd_f32")
    return g


class LpgFunction(torch.autograd.Function):
    """autograd shell over the two native LPG kernels (the reference pairs LocalPlanarGuidance with a registered
    LocalPlanarGuidanceGrad, tensorflow/custom_layer/_local_planar_guidance_grad.py:22-33)."""

    @staticmethod
    def forward(ctx, plane_eq, upratio, abs_min):
        ctx.save_for_backward(plane_eq)
        ctx.upratio = int(upratio)
        return lpg_forward(plane_eq.detach(), upratio, abs_min=abs_min)

    @staticmethod
    def backward(ctx, grad_depth):
        (plane_eq,) = ctx.saved_tensors
        return lpg_backward(plane_eq, grad_depth, ctx.upratio), None, None


def lpg_fused_forward(plane4: torch.Tensor, B: int, h: int, w: int, upratio: int, max_depth: float,
                      normalize: bool, depth_scaled: torch.Tensor, ds_out: Optional[torch.Tensor] = None,
                      ds_factor: int = 1, ds_pix_stride: int = 1, abs_min: Optional[torch.Tensor] = None):
    """LPG + glue of bts.forward (reference bts.py:250-256): plane4 [B*h*w,4] -> depth/max_depth."""
    _need(plane4, "lpg_fused_forward")
    _need(depth_scaled, "lpg_fused_forward")
    k = int(upratio)
    if plane4.numel() != B * h * w * 4 or not plane4.is_contiguous():
        raise BtsHipError("lpg_fused_forward: plane4 must be contiguous [B*h*w,4]")
    if depth_scaled.numel() != B * h * k * w * k or not depth_scaled.is_contiguous():
        raise BtsHipError("lpg_fused_forward: depth_scaled must be contiguous [B,1,h*k,w*k]")
    nbytes = 4.0 * (4 * B * h * w + B * h * k * w * k + (B * h * k * w * k // (ds_factor * ds_factor) if ds_out is not None else 0))
    with torch.cuda.device(plane4.device):
        rc = _launch("lpg_fwd_kernel<%d,fused>" % k, "lpg", 8.0 * B * h * k * w * k, nbytes,
                     lambda: _lib.load().bts_lpg_fused_fwd_f32(_ptr(plane4), B, h, w, k, int(bool(normalize)),
                                                               float(max_depth), _ptr(depth_scaled), _ptr(ds_out),
                                                               int(ds_factor), int(ds_pix_stride), _ptr(abs_min),
                                                               _stream(plane4)))
    _lib.check(rc, "bts_lpg_fused_fwd_f32")
    return depth_scaled


# ------------------------------------------------------------------------ reduction
def reduc_chain(num_in: int, num_out: int) -> List[Tuple[int, int]]:
    """(cin, cout_real) per layer, mirroring the while-loop of reduction_1x1.__init__ (bts.py:105-122).
    The last entry's cout is filled by the caller's weights (3 or 1)."""
    layers = []
    while num_out >= 4:
        if num_out < 8:
            layers.append((num_in, -1))
            break
        layers.append((num_in, num_out))
        num_in, num_out = num_out, num_out // 2
    return layers


def reduc_uses_mfma16(c_in: int, c_first_out: int) -> bool:
    """The narrow chains (bts_size 512: 2x2 64->32.., 1x1 32->16..; every chain of bts_size 256) run on the 16x16x4-MFMA kernel
    (csrc/reduc.hip)."""
    return (c_in, c_first_out) in ((64, 32), (32, 16), (64, 64), (16, 8))      # the last two: bts_size 256


def pack_reduc_weights(weights: Sequence[torch.Tensor]) -> torch.Tensor:
    """Pack a reduction chain's 1x1 weights ([cout,cin,1,1] each) into MFMA fragment order.

    Wide chains (first layer 128 -> ..): per layer (K=cin, rows padded to 32*MT) float4 index
    ((mt*(K/8)+g)*64 + 32*h + i) holds W[32*mt+i][4*(2g+h) + 0..3] -- lane (i,h) of v_mfma_f32_32x32x2_f32's A operand
    for the four k-steps of group g.  Narrow chains (see reduc_uses_mfma16): rows padded to 16*MT, K to 16*G, float4
    index ((mt*G+g)*64 + 16*kq + i) holds W[16*mt+i][16*g + 4*kq + 0..3] -- lane (i,kq) of v_mfma_f32_16x16x4_f32."""
    c_in = weights[0].shape[1]
    c_first = weights[0].shape[0]
    narrow = reduc_uses_mfma16(c_in, c_first)
    parts = []
    for w in weights:
        cout, cin = w.shape[0], w.shape[1]
        assert cin % 8 == 0, "reduction chain widths are multiples of 8"
        if narrow:
            mt, g = (cout + 15) // 16, (cin + 15) // 16
            wp = torch.zeros((mt * 16, g * 16), dtype=torch.float32, device=w.device)
            wp[:cout, :cin] = w.reshape(cout, cin).float()
            # (mt, i, g, kq, q4) -> (mt, g, kq, i, q4)
            parts.append(wp.view(mt, 16, g, 4, 4).permute(0, 2, 3, 1, 4).contiguous().view(-1))
        else:
            mt = (cout + 31) // 32
            wp = torch.zeros((mt * 32, cin), dtype=torch.float32, device=w.device)
            wp[:cout] = w.reshape(cout, cin).float()
            # (mt, i, g, h, q) -> (mt, g, h, i, q)
            parts.append(wp.view(mt, 32, cin // 8, 2, 4).permute(0, 2, 3, 1, 4).contiguous().view(-1))
    return torch.cat(parts).contiguous()


def reduc_forward_nhwc(x2d: torch.Tensor, c_in: int, c_first_out: int, w_frag: torch.Tensor, max_depth: float,
                       is_final: bool, normalize: bool, out: torch.Tensor):
    """x2d: [npix, >=c_in] NHWC view; out: [npix,4] (non-final) or [npix] (final), contiguous."""
    stride, cview = _rows2d(x2d, "reduc_forward_nhwc")
    _need(out, "reduc_forward_nhwc")
    _need(w_frag, "reduc_forward_nhwc")
    if cview < c_in:
        raise BtsHipError("reduc_forward_nhwc: view has %d channels, chain needs %d" % (cview, c_in))
    npix = x2d.shape[0]
    if out.numel() != npix * (1 if is_final else 4) or not out.is_contiguous():
        raise BtsHipError("reduc_forward_nhwc: bad output size")
    chain = reduc_chain(c_in, c_first_out)
    macs = sum(ci * (co if co > 0 else (1 if is_final else 3)) for ci, co in chain)
    nbytes = 4.0 * (npix * (c_in + (1 if is_final else 4)) + macs)
    tops = torch_ops()
    if tops is not None:
        run = lambda: _op(lambda: tops.reduction_1x1(x2d, int(c_in), int(c_first_out), w_frag, float(max_depth), bool(is_final),
                                                     bool(normalize), out))
    else:
        run = lambda: _lib.load().bts_reduc_fwd_f32(_ptr(x2d), stride, npix, int(c_in), int(c_first_out),
                                                    _ptr(w_frag), w_frag.numel(), float(max_depth),
                                                    int(bool(is_final)), int(bool(normalize)), _ptr(out), _stream(x2d))
    with torch.cuda.device(x2d.device):
        rc = _launch("reduc_fwd_kernel<%d,%d>" % (c_in, c_first_out), "reduc", 2.0 * npix * macs, nbytes, run)
    _lib.check(rc, "bts_reduc_fwd_f32")
    return out


def reduc_lpg_forward(x2d: torch.Tensor, B: int, h: int, w: int, c_in: int, c_first_out: int, w_frag: torch.Tensor,
                      max_depth: float, upratio: int, depth_scaled: torch.Tensor, ds_out: Optional[torch.Tensor] = None,
                      abs_min: Optional[torch.Tensor] = None, plane4: Optional[torch.Tensor] = None):
    """One scale of the decoder's LPG stage in ONE launch (bts_reduc_lpg_fwd_f32): reduction_1x1 chain -> normalize ->
    LPG -> /max_depth (+ the nearest-downsampled plane, + abs_min).  x2d: [B*h*w, >=c_in] NHWC view; depth_scaled:
    contiguous [B,1,h*k,w*k]; ds_out: contiguous [B*2h*2w] plane (k = 8, 4) or None."""
    stride, cview = _rows2d(x2d, "reduc_lpg_forward")
    _need(depth_scaled, "reduc_lpg_forward")
    _need(w_frag, "reduc_lpg_forward")
    k = int(upratio)
    npix = B * h * w
    if cview < c_in or x2d.shape[0] != npix:
        raise BtsHipError("reduc_lpg_forward: bad input view %s for B=%d %dx%d, %d channels" % (tuple(x2d.shape), B, h, w, c_in))
    if depth_scaled.numel() != npix * k * k or not depth_scaled.is_contiguous():
        raise BtsHipError("reduc_lpg_forward: depth_scaled must be contiguous [B,1,h*k,w*k]")
    if ds_out is not None:
        _need(ds_out, "reduc_lpg_forward")
        if k == 2 or ds_out.numel() != npix * 4 or not ds_out.is_contiguous():
            raise BtsHipError("reduc_lpg_forward: ds_out must be a contiguous [B,2h,2w] plane (k = 8 or 4 only)")
    if plane4 is not None and (plane4.numel() != npix * 4 or not plane4.is_contiguous()):
        raise BtsHipError("reduc_lpg_forward: plane4 must be contiguous [B*h*w,4]")
    chain = reduc_chain(c_in, c_first_out)
    macs = sum(ci * (co if co > 0 else 3) for ci, co in chain)
    nbytes = 4.0 * (npix * c_in + macs + npix * k * k + (npix * 4 if ds_out is not None else 0))
    tops = torch_ops()
    if tops is not None:
        run = lambda: _op(lambda: tops.reduc_lpg(x2d, B, h, w, int(c_in), int(c_first_out), w_frag, float(max_depth), k,
                                                 depth_scaled, ds_out, abs_min, plane4))
    else:
        run = lambda: _lib.load().bts_reduc_lpg_fwd_f32(_ptr(x2d), stride, B, h, w, int(c_in), int(c_first_out),
                                                        _ptr(w_frag), w_frag.numel(), float(max_depth), k,
                                                        _ptr(plane4), _ptr(depth_scaled), _ptr(ds_out),
                                                        _ptr(abs_min), _stream(x2d))
    with torch.cuda.device(x2d.device):
        rc = _launch("reduc_lpg_kernel<%d,%d,k%d>" % (c_in, c_first_out, k), "reduc_lpg", 2.0 * npix * macs + 8.0 * npix * k * k, nbytes, run)
    _lib.check(rc, "bts_reduc_lpg_fwd_f32")
    return depth_scaled


# ---------------------------------------------------------------------------- layout
def nchw_to_nhwc(src: torch.Tensor, dst2d: torch.Tensor, relu: bool = False):
    """src [B,C,H,W] contiguous -> dst2d [B*H*W, C] view (channel slice of an NHWC buffer)."""
    _need(src, "nchw_to_nhwc")
    stride, cview = _rows2d(dst2d, "nchw_to_nhwc")
    src = src.contiguous()
    B, Cc, H, W = src.shape
    if cview != Cc or dst2d.shape[0] != B * H * W:
        raise BtsHipError("nchw_to_nhwc: destination view shape mismatch")
    with torch.cuda.device(src.device):
        rc = _launch("nchw_to_nhwc_kernel", "layout", 0.0, 8.0 * src.numel(),
                     lambda: _lib.load().bts_nchw_to_nhwc_f32(_ptr(src), B, Cc, H * W, _ptr(dst2d), stride, int(bool(relu)), _stream(src)))
    _lib.check(rc, "bts_nchw_to_nhwc_f32")
    return dst2d


def nhwc_to_nchw(src2d: torch.Tensor, B: int, H: int, W: int) -> torch.Tensor:
    stride, Cc = _rows2d(src2d, "nhwc_to_nchw")
    if src2d.shape[0] != B * H * W:
        raise BtsHipError("nhwc_to_nchw: source view shape mismatch")
    dst = torch.empty((B, Cc, H, W), dtype=torch.float32, device=src2d.device)
    with torch.cuda.device(src2d.device):
        rc = _lib.load().bts_nhwc_to_nchw_f32(_ptr(src2d), stride, B, Cc, H * W, _ptr(dst), _stream(src2d))
    _lib.check(rc, "bts_nhwc_to_nchw_f32")
    return dst


# ------------------------------------------------------------------------------ conv
def round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def pack_conv_weight(w: torch.Tensor, perm: Optional[torch.Tensor] = None,
                     c_in_ld: Optional[int] = None) -> Tuple[torch.Tensor, int, int]:
    """[cout,cin,k,k] -> packed [cout_pad][k_pad], K flattened tap-major: k = tap*c_in_ld + c, zero padded
    (c_in_ld = channels the kernel walks per tap, a multiple of 4 >= cin; k_pad = K rounded up to 32).
    ``perm``: optional LongTensor; buffer channel j reads reference input channel perm[j]
    (lets a concat buffer keep its own channel order).  Returns (packed, cout_pad, c_in_ld)."""
    cout, cin, kh, kw = w.shape
    wf = w.float()
    if perm is not None:
        wf = wf[:, perm.to(w.device)]
    if c_in_ld is None:
        c_in_ld = round_up(cin, 4)
    assert c_in_ld % 4 == 0 and c_in_ld >= cin
    cout_pad = round_up(cout, 32)
    k_flat = kh * kw * c_in_ld
    k_pad = round_up(k_flat, 32)
    p = torch.zeros((cout_pad, kh * kw, c_in_ld), dtype=torch.float32, device=w.device)
    p[:cout, :, :cin] = wf.permute(0, 2, 3, 1).reshape(cout, kh * kw, cin)
    out = torch.zeros((cout_pad, k_pad), dtype=torch.float32, device=w.device)
    out[:, :k_flat] = p.reshape(cout_pad, k_flat)
    return out.contiguous(), cout_pad, c_in_ld


def pack_grouped_conv_weight(w: torch.Tensor, groups: int) -> Tuple[torch.Tensor, int, int]:
    """Grouped [c, c/groups, k, k] weight (cin == cout == c, ResNeXt's 3x3) -> [n_bundles, cb, k_pad] for
    conv_forward(n_bundles=...): consecutive groups are packed block-diagonally into bundles of cb = max(32, c/groups)
    channels, so each bundle is an ordinary dense convolution over its own cb input channels.
    Returns (packed, n_bundles, cb)."""
    c, cg, kh, kw = w.shape
    if c != cg * groups:
        raise BtsHipError("pack_grouped_conv_weight: expected cin == cout == groups * channels-per-group")
    cb = max(32, cg)
    if cb % cg or c % cb or cb % 32:
        raise BtsHipError("pack_grouped_conv_weight: unsupported group width %d" % cg)
    nb, gpb = c // cb, cb // cg                             # bundles, groups per bundle
    dense = torch.zeros((nb, cb, cb, kh, kw), dtype=torch.float32, device=w.device)
    wv = w.float().reshape(nb, gpb, cg, cg, kh, kw)         # [bundle, group-in-bundle, out-in-group, in-in-group, k, k]
    for g in range(gpb):
        dense[:, g * cg:(g + 1) * cg, g * cg:(g + 1) * cg] = wv[:, g]
    packed = torch.stack([pack_conv_weight(dense[j], c_in_ld=cb)[0] for j in range(nb)])
    return packed.contiguous(), nb, cb


_SUBPIX_SETS = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}   # (parity, tap) -> 3x3 kernel rows summed


def pack_upconv_subpixel(w: torch.Tensor, c_in_ld: Optional[int] = None) -> Tuple[torch.Tensor, int, int]:
    """upconv = nearest-2x + conv3x3 (reference bts.py:90-92) as four 2x2 convolutions on the source.

    Output pixel (2Y+py, 2X+px) only ever sees source rows {Y-1+py, Y+py} and columns {X-1+px, X+px}: the
    three kernel rows collapse onto two source rows (py=0: {k0 | k1+k2}, py=1: {k0+k1 | k2}), same for
    columns.  Returns ([4][cout_pad][k_pad] with class = 2*py+px and k = (ty*2+tx)*c_in_ld + c, cout_pad,
    c_in_ld).  Zero padding of the upsampled map coincides with source out-of-range, so borders are exact."""
    cout, cin, kh, kw = w.shape
    assert kh == 3 and kw == 3
    if c_in_ld is None:
        c_in_ld = round_up(cin, 4)
    cout_pad = round_up(cout, 32)
    k_pad = round_up(4 * c_in_ld, 32)
    wf = w.float()
    out = torch.zeros((4, cout_pad, k_pad), dtype=torch.float32, device=w.device)
    for py in (0, 1):
        for px in (0, 1):
            blk = torch.zeros((cout_pad, 4, c_in_ld), dtype=torch.float32, device=w.device)
            for ty in (0, 1):
                for tx in (0, 1):
                    acc = None
                    for ky in _SUBPIX_SETS[(py, ty)]:
                        for kx in _SUBPIX_SETS[(px, tx)]:
                            acc = wf[:, :, ky, kx] if acc is None else acc + wf[:, :, ky, kx]
                    blk[:cout, ty * 2 + tx, :cin] = acc
            out[2 * py + px, :, :4 * c_in_ld] = blk.reshape(cout_pad, 4 * c_in_ld)
    return out.contiguous(), cout_pad, c_in_ld


def pack_upconv_taps(w: torch.Tensor, c_in_ld: Optional[int] = None) -> Tuple[torch.Tensor, int, int]:
    """upconv as a TAP GEMM (bts_upconv_combine_f32): the 3x3 kernel [cout, cin, 3, 3] as the weight of ONE 1x1
    convolution with 9*cout outputs, row t*cout + n = w[n, :, ky, kx] with t = 3*ky + kx.  Returns
    ([9*cout (padded to 32)][k_pad], 9*cout, c_in_ld); cout must be a multiple of 4."""
    cout, cin, kh, kw = w.shape
    assert kh == 3 and kw == 3 and cout % 4 == 0
    if c_in_ld is None:
        c_in_ld = round_up(cin, 4)
    rows = 9 * cout
    out = torch.zeros((round_up(rows, 32), round_up(c_in_ld, 32)), dtype=torch.float32, device=w.device)
    out[:rows, :cin] = w.float().permute(2, 3, 0, 1).reshape(rows, cin)
    return out.contiguous(), rows, c_in_ld


def upconv_combine(taps2d: torch.Tensor, B: int, h: int, w: int, c: int, y2d: torch.Tensor, act: int = ACT_NONE,
                   e2: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, tag: str = "decoder_upconv_sum"):
    """Second stage of the tap-GEMM upconv (bts_upconv_combine_f32): taps2d [B*h*w, >= 9*c] -> y2d [B*2h*2w, c] NHWC view."""
    ts, tc = _rows2d(taps2d, "upconv_combine")
    ys, yc = _rows2d(y2d, "upconv_combine")
    if tc < 9 * c or yc != c or c % 4 or taps2d.shape[0] != B * h * w or y2d.shape[0] != 4 * B * h * w:
        raise BtsHipError("upconv_combine: views %s / %s do not fit B=%d %dx%d c=%d"
                          % (tuple(taps2d.shape), tuple(y2d.shape), B, h, w, c))
    es = eb = None
    if e2 is not None:
        es, eb = e2
        _need(es, "upconv_combine")
        _need(eb, "upconv_combine")
        if es.numel() < c or eb.numel() < c:
            raise BtsHipError("upconv_combine: e2 vectors need %d elements" % c)
    npx = B * h * w
    with torch.cuda.device(taps2d.device):
        rc = _launch("upconv_combine_kernel", tag, 0.0, 4.0 * (9 * npx * c + 4 * npx * c),
                     lambda: _lib.load().bts_upconv_combine_f32(_ptr(taps2d), ts, B, h, w, c, _ptr(es), _ptr(eb), int(act),
                                                                _ptr(y2d), ys, _stream(taps2d)))
    _lib.check(rc, "bts_upconv_combine_f32")
    return y2d


def pad_vec(v: Optional[torch.Tensor], n: int, fill: float = 0.0) -> Optional[torch.Tensor]:
    if v is None:
        return None
    out = torch.full((n,), fill, dtype=torch.float32, device=v.device)
    out[: v.numel()] = v.float().reshape(-1)
    return out


def bn_affine(weight, bias, mean, var, eps: float):
    """Eval-mode BatchNorm as y = x*scale + shift (ATen batch_norm_cpu_transform_input form)."""
    invstd = 1.0 / torch.sqrt(var.float() + eps)
    scale = weight.float() * invstd
    shift = bias.float() - mean.float() * scale
    return scale, shift


# ---- per-call launch configuration -----------------------------------------------------------------------------------
# Two facts about a convolution launch belong to the CALLER, not to the process: the arithmetic of the contraction
# (bts_conv_desc.precision: 0 = fp32-input MFMA, 1 = fp32 emulated on the bf16 matrix cores) and the number of frames
# the caller expects to share the chip (bts_conv_desc.fill_frames, which sizes split-K and the tile family; 0 = the
# library default of 8).  Both change output BITS (fp32 summation order / product rounding), so two models in one
# process must be able to hold different values: they live in a thread-local scope that a model opens around its own
# forward (``BtsModel.fill_frames`` / ``BtsModel.conv_precision``, bts_amd/bts.py) -- there is no process-wide setter.
# Code that calls conv_forward directly (tests, micro-benchmarks) wraps the calls in ``launch_config(...)``.
_cfg_tls = threading.local()
_PRECISIONS = {"fp32": 0, "bf16x3": 1, 0: 0, 1: 1}


class launch_config:
    """``with ops.launch_config(fill_frames=2, precision="bf16x3"): ...`` -- values for every conv_forward call issued by
    this thread inside the block; ``None`` keeps the enclosing value (library defaults outside any block)."""

    def __init__(self, fill_frames: Optional[int] = None, precision=None):
        if fill_frames is not None:
            fill_frames = int(fill_frames)
            if fill_frames < 0 or fill_frames > 4096:
                raise BtsHipError("launch_config: fill_frames must be in 0..4096")
        if precision is not None:
            if precision not in _PRECISIONS:
                raise BtsHipError("launch_config: precision must be 'fp32' / 0 or 'bf16x3' / 1")
            precision = _PRECISIONS[precision]
        self._new = (fill_frames, precision)

    def __enter__(self):
        self._prev_raw = getattr(_cfg_tls, "value", None)
        self._prev = current_launch_config()
        ff, pr = self._new
        _cfg_tls.value = (self._prev[0] if ff is None else ff, self._prev[1] if pr is None else pr)
        return self

    def __exit__(self, *exc):
        _cfg_tls.value = self._prev_raw
        return False


# Outside any launch_config scope the precision is $BTS_CONV_PRECISION (0 / 1, the knob libbts_hip.so itself reads as an
# override of every descriptor): `BTS_CONV_PRECISION=1 python -m pytest tests -m gpu` runs the WHOLE suite -- module-level
# ops, models, training forward AND backward (autograd runs the backward outside the model's scope) -- in the emulated
# arithmetic, pre-split weights included.
_ENV_PRECISION = 1 if os.environ.get("BTS_CONV_PRECISION", "0").strip() == "1" else 0


def current_launch_config() -> Tuple[int, int]:
    """(fill_frames, precision) in force for this thread."""
    v = getattr(_cfg_tls, "value", None)
    if v is None:
        return (0, _ENV_PRECISION)
    return (v[0], 1 if _ENV_PRECISION else v[1])


def launch_config_active() -> bool:
    """True inside a ``launch_config`` block of this thread (a model called from another model's forward keeps the
    outer declaration)."""
    return getattr(_cfg_tls, "value", None) is not None


def model_launch_config(module, batch: int) -> "launch_config":
    """The scope a model opens around its forward: its ``fill_frames`` attribute (None = ``auto_fill_frames(batch)``)
    and its ``conv_precision`` attribute."""
    ff = getattr(module, "fill_frames", None)
    return launch_config(auto_fill_frames(batch) if ff is None else ff, getattr(module, "conv_precision", 0))


def auto_fill_frames(batch: int) -> int:
    """The frames-per-launch declaration a model makes when its ``fill_frames`` attribute is None: a function of the
    batch it was called with, in three coarse classes (so bits change only between classes, documented in DESIGN 5a):
    single-frame callers (the reference's own test loop, bts_test.py:127-147: B = 1) get the latency setting; small
    batches the library default; chip-filling batches the setting bench.py declares."""
    if batch <= 2:
        return 2
    if batch <= 11:
        return 8
    return 16


def split_bf16x3(w: torch.Tensor) -> torch.Tensor:
    """The emulated mode's three-way truncation split of a float32 tensor, done offline: returns int16 [.., 3, R, K] for a
    [.., R, K] input -- plane 0 = top 16 bits of w, plane 1 = top 16 bits of (w - plane 0), plane 2 = top 16 bits of the
    remainder (both subtractions are exact in fp32; w - (h + m + l) <= 2^-24 |w|).  Bit for bit what the kernels' own
    split_store does to an operand on its way to LDS (csrc/conv_mfma.hip)."""
    _need(w, "split_bf16x3")
    hi = w.view(torch.int32) & -65536
    r1 = w - hi.view(torch.float32)
    mid = r1.view(torch.int32) & -65536
    lo = (r1 - mid.view(torch.float32)).view(torch.int32) & -65536
    planes = torch.stack([hi, mid, lo], dim=-3)
    return (planes >> 16).to(torch.int16).contiguous()


_WINO = os.environ.get("BTS_CONV_WINO", "1").strip() not in ("", "0")      # fused Winograd F(2x2,3x3) for eligible 3x3 layers (0: direct kernels, A/B)


def pack_wino_weight(w_packed: torch.Tensor, c_in_ld: int, n_tail: int = 0, c_out16: int = 0,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Winograd F(2x2,3x3) form of a packed 3x3 weight ([c_out_pad, round_up(9 * c_in_ld, 32)], tap-major K as
    pack_conv_weight lays it out), in the B-fragment order conv_wino_kernel loads: one launch of bts_pack_wino_f32
    (include/bts_hip.h documents the layout; pack_wino_weight_reference is the torch statement of the same).  ``out``:
    refill an existing buffer (the training step re-transforms its weights every iteration)."""
    _need(w_packed, "pack_wino_weight")
    cop = w_packed.shape[0]
    if w_packed.dim() != 2 or not w_packed.is_contiguous() or w_packed.shape[1] != round_up(9 * c_in_ld, 32):
        raise BtsHipError("pack_wino_weight: needs a packed 3x3 weight [c_out_pad, round_up(9 * c_in_ld, 32)]")
    lib = _lib.load_real()
    n = lib.bts_pack_wino_floats(cop, c_in_ld, n_tail, c_out16)
    if n <= 0:
        raise BtsHipError("pack_wino_weight: needs a 3x3 weight whose buffer channels are whole 32-channel chunks"
                          " (and c_out16 a multiple of 16 within the packed rows)")
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=w_packed.device)
    elif out.numel() != n or not out.is_contiguous() or out.device != w_packed.device:
        raise BtsHipError("pack_wino_weight: `out` does not fit this weight")
    with torch.cuda.device(w_packed.device):
        _lib.check(lib.bts_pack_wino_f32(w_packed.data_ptr(), cop, w_packed.shape[1], c_in_ld, n_tail, c_out16, out.data_ptr(),
                                         _stream(w_packed)), "bts_pack_wino_f32")
    return out


def pack_wino_weight_reference(w_packed: torch.Tensor, c_in_ld: int, n_tail: int = 0, c_out16: int = 0) -> torch.Tensor:
    """torch (fp64 einsum) statement of pack_wino_weight, kept as its test oracle.  Winograd F(2x2,3x3) form of a packed 3x3 weight ([c_out_pad, 9 * c_in_ld], tap-major K as pack_conv_weight lays
    it out): U = G g G^T per (output, input) channel, computed in fp64 and rounded once, in the B-fragment order
    conv_wino_kernel loads.  Default (32-wide channel tiles, v_mfma_f32_32x32x2_f32): float index
    (((((xi * nchunks + chunk) * n_ct + ct) * 4 + g) * 64 + lh * 32 + li) * 4 + q = U[xi][n = 32 ct + li][k = 32 chunk + 8 g + 4 lh + q].
    ``c_out16`` > 0 (the 48-wide tile, v_mfma_f32_16x16x4_f32; c_out16 = real output channels, a multiple of 16):
    (((((xi * nchunks + chunk) * n_ct16 + ct) * 2 + g) * 64 + l) * 4 + q = U[xi][n = 16 ct + (l & 15)][k = 32 chunk + 16 g + 4 (l >> 4) + q].
    ``n_tail`` > 0: the last 4 of the c_in_ld input channels are the planar tail operand (bts_conv_desc.tail_planes): U
    covers the c_in_ld - 4 buffer channels only (whole chunks) -- the kernel adds the tail's products directly from the
    packed fp32 weights."""
    _need(w_packed, "pack_wino_weight")
    cop = w_packed.shape[0]
    c_main = c_in_ld - (4 if n_tail else 0)
    if c_main <= 0 or c_main % 32 or cop % 32 or w_packed.shape[1] != round_up(9 * c_in_ld, 32):
        raise BtsHipError("pack_wino_weight: needs a 3x3 weight whose buffer channels are whole 32-channel chunks")
    g = w_packed[:, :9 * c_in_ld].reshape(cop, 3, 3, c_in_ld).permute(0, 3, 1, 2).double()     # [n, k, 3, 3]
    g = g[:, :c_main]
    G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64, device=w_packed.device)
    U = torch.einsum("ia,nkab,jb->nkij", G, g, G).float().reshape(cop, c_main, 16)              # [n, k, xi = 4 i + j]
    nchunks = c_main // 32
    if c_out16:
        if c_out16 % 16 or c_out16 > cop:
            raise BtsHipError("pack_wino_weight: c_out16 must be a multiple of 16 within the packed rows")
        U = U[:c_out16].reshape(c_out16 // 16, 16, nchunks, 2, 4, 4, 16)                        # (ct, col, chunk, g, kq, q, xi)
        return U.permute(6, 2, 0, 3, 4, 1, 5).contiguous().view(-1)                              # (xi, chunk, ct, g, kq, col, q): lane = 16 kq + col
    U = U.view(cop // 32, 32, nchunks, 4, 2, 4, 16)                                             # (ct, li, chunk, g, lh, q, xi)
    return U.permute(6, 2, 0, 3, 4, 1, 5).contiguous().view(-1)                                  # (xi, chunk, ct, g, lh, li, q)


def conv_forward(x2d: torch.Tensor, B: int, h_in: int, w_in: int, w_packed: torch.Tensor, c_out: int,
                 ksize: int, dil: int = 1, up: int = 1, c_in_ld: Optional[int] = None,
                 pre: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, pre_relu: bool = False,
                 e1: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, act: int = ACT_NONE,
                 e2: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                 y2d: Optional[torch.Tensor] = None, y_nchw: Optional[torch.Tensor] = None,
                 tag: str = "conv", c_in_real: Optional[int] = None, stride: int = 1, pad: Optional[int] = None,
                 y2_2d: Optional[torch.Tensor] = None, subpixel: bool = False,
                 splitk_ws: Optional[torch.Tensor] = None, res2d: Optional[torch.Tensor] = None, n_bundles: int = 1,
                 tail_planes: Optional[Sequence[torch.Tensor]] = None, algo_flops: Optional[float] = None):
    """One fused convolution (see bts_conv_desc in include/bts_hip.h).
    ``algo_flops``: FLOPs of the reference formulation this launch stands for, when that is not the launch's own count
    (the tap GEMM of an upconv: 9 tap-products per source pixel for the reference's 36); only used by KernelTrace.
    ``tail_planes``: 1..4 contiguous one-channel maps ([B,1,h_in,w_in] or [B,h_in,w_in]) that supply the LAST input
    channels of the reference's concatenated input (bts.py:260, 274, 287) without ever being copied into the NHWC
    buffer: x2d then holds c_in_ld - 4 channels and w_packed is packed with c_in_ld = (buffer channels) + 4.
    ``subpixel``: w_packed comes from pack_upconv_subpixel; computes nearest-2x + conv3x3 (pass ksize=3, up=2).

    x2d: [B*h_in*w_in, C>=c_in_ld] NHWC view.  Exactly one of y2d ([B*H*W, c_out] NHWC view) /
    y_nchw ([B,c_out,H,W] contiguous) receives the result.  pad defaults to dil*(ksize//2).
    ``res2d``: residual [B*H*W, c_out] added after e1, before the activation.  ``n_bundles`` > 1: grouped convolution
    as independent channel bundles (w_packed [n_bundles, c_out_pad, k_pad] from pack_grouped_conv_weight; c_in_ld /
    c_out are PER BUNDLE, x2d / y2d hold all n_bundles*c_in_ld / n_bundles*c_out channels)."""
    xs, xc = _rows2d(x2d, "conv_forward")
    _need(w_packed, "conv_forward")
    n_tail = len(tail_planes) if tail_planes else 0
    if c_in_ld is None:
        c_in_ld = xc + (4 if n_tail else 0)
    flops_taps = ksize * ksize
    if subpixel:
        if ksize != 3 or up != 2 or dil != 1 or stride != 1 or w_packed.dim() != 3 or w_packed.shape[0] != 4:
            raise BtsHipError("conv_forward: subpixel needs ksize=3, up=2 and weights from pack_upconv_subpixel")
        _, c_out_pad, k_pad = w_packed.shape
        taps = 4
    elif n_bundles > 1:
        if w_packed.dim() != 3 or w_packed.shape[0] != n_bundles or c_in_ld is None:
            raise BtsHipError("conv_forward: bundled weights must be [n_bundles, c_out_pad, k_pad] with c_in_ld per bundle")
        _, c_out_pad, k_pad = w_packed.shape
        taps = ksize * ksize
    else:
        c_out_pad, k_pad = w_packed.shape
        taps = ksize * ksize
    if k_pad != round_up(taps * c_in_ld, 32) or not w_packed.is_contiguous():
        raise BtsHipError("conv_forward: packed weight [%d,%d] does not match ksize %d / c_in_ld %d"
                          % (c_out_pad, k_pad, ksize, c_in_ld))
    if c_in_ld % 4 or (c_in_ld - (4 if n_tail else 0)) * n_bundles > xc or x2d.shape[0] != B * h_in * w_in:
        raise BtsHipError("conv_forward: bad input view (c_in_ld %d, view %s)" % (c_in_ld, tuple(x2d.shape)))
    if n_tail:
        if n_tail > 4 or ksize != 3 or stride != 1 or dil != 1 or up != 1 or subpixel or n_bundles > 1:
            raise BtsHipError("conv_forward: tail_planes need a plain 3x3 / stride 1 / dilation 1 convolution and at most 4 planes")
        for t in tail_planes:
            _need(t, "conv_forward")
            if t.numel() != B * h_in * w_in or not t.is_contiguous():
                raise BtsHipError("conv_forward: every tail plane must be a contiguous [B,1,h_in,w_in] map")
    if pad is None:
        pad = dil * (ksize // 2)
    H = (h_in * up + 2 * pad - dil * (ksize - 1) - 1) // stride + 1
    W = (w_in * up + 2 * pad - dil * (ksize - 1) - 1) // stride + 1
    d = ConvDesc()
    d.x, d.x_pix_stride, d.c_in_ld, d.k_pad = x2d.data_ptr(), xs, c_in_ld, k_pad
    d.B, d.h_in, d.w_in, d.up, d.ksize, d.dil, d.stride, d.pad = B, h_in, w_in, up, ksize, dil, stride, pad
    if subpixel:
        d.up, d.ksize, d.pad, d.subpixel = 1, 2, 0, 1
    d.w, d.c_out, d.c_out_pad = w_packed.data_ptr(), c_out, c_out_pad
    keep = []
    d.n_bundles = n_bundles if n_bundles > 1 else 0
    d.fill_frames, d.precision = current_launch_config()
    if d.precision == 1 and n_bundles <= 1 and not n_tail:
        # weights pre-split into bf16 planes for the emulated mode's halo-tile kernel (LDS-DMA of plain bytes); made once
        # per packed weight tensor and kept on it
        ws3 = getattr(w_packed, "_bts_split3", None)
        seq3 = getattr(w_packed, "_bts_pack_seq", 0)       # train.WeightPacker refills packed buffers in place: re-split then
        if ws3 is None or getattr(w_packed, "_bts_split3_seq", 0) != seq3:
            if ws3 is None:
                ws3 = split_bf16x3(w_packed)
            else:
                ws3.copy_(split_bf16x3(w_packed))
            w_packed._bts_split3, w_packed._bts_split3_seq = ws3, seq3
        keep.append(ws3)
        d.w_split = ws3.data_ptr()
    d.n_tail = n_tail
    for j in range(n_tail):
        d.tail_planes[j] = tail_planes[j].data_ptr()
    for name, pair, n in (("pre", pre, c_in_ld * n_bundles), ("e1", e1, c_out_pad * n_bundles), ("e2", e2, c_out_pad * n_bundles)):
        if pair is not None:
            s, b = pair
            if s.numel() != n or b.numel() != n:
                raise BtsHipError("conv_forward: %s vectors must have %d elements" % (name, n))
            _need(s, "conv_forward")
            _need(b, "conv_forward")
            keep += [s, b]
            setattr(d, name + "_scale", s.data_ptr())
            setattr(d, name + "_shift", b.data_ptr())
    d.pre_relu, d.act = int(bool(pre_relu)), int(act)
    if (y2d is None) == (y_nchw is None):
        raise BtsHipError("conv_forward: give exactly one of y2d / y_nchw")
    if y2d is not None:
        ys, yc = _rows2d(y2d, "conv_forward")
        if yc != c_out * n_bundles or y2d.shape[0] != B * H * W:
            raise BtsHipError("conv_forward: bad output view")
        d.y, d.y_pix_stride, d.y_nchw = y2d.data_ptr(), ys, 0
        out = y2d
        if y2_2d is not None:
            y2s, y2c = _rows2d(y2_2d, "conv_forward")
            if y2c != c_out * n_bundles or y2_2d.shape[0] != B * H * W:
                raise BtsHipError("conv_forward: bad second output view")
            d.y2, d.y2_pix_stride = y2_2d.data_ptr(), y2s
        if res2d is not None:
            rs, rc = _rows2d(res2d, "conv_forward")
            if rc != c_out * n_bundles or res2d.shape[0] != B * H * W:
                raise BtsHipError("conv_forward: bad residual view %s" % (tuple(res2d.shape),))
            d.res, d.res_pix_stride = res2d.data_ptr(), rs
    else:
        if res2d is not None or n_bundles > 1:
            raise BtsHipError("conv_forward: residual / bundled convolutions write NHWC only")
        _need(y_nchw, "conv_forward")
        if tuple(y_nchw.shape) != (B, c_out, H, W) or not y_nchw.is_contiguous():
            raise BtsHipError("conv_forward: y_nchw must be contiguous [B,c_out,H,W]")
        d.y, d.y_pix_stride, d.y_nchw = y_nchw.data_ptr(), 0, 1
        out = y_nchw
    if splitk_ws is not None:
        _need(splitk_ws, "conv_forward")
        if not splitk_ws.is_contiguous():
            raise BtsHipError("conv_forward: splitk_ws must be contiguous")
        d.splitk_ws, d.splitk_ws_floats = splitk_ws.data_ptr(), splitk_ws.numel()
    if (_WINO and d.precision == 0 and ksize == 3 and stride == 1 and dil == 1 and pad == 1 and up == 1 and not subpixel
            and n_bundles <= 1 and (c_in_ld - (4 if n_tail else 0)) % 32 == 0 and c_in_ld > 4 and y_nchw is None):
        # Winograd-form weights for the fused F(2x2,3x3) kernel: made once per packed weight and kept on it
        uw = getattr(w_packed, "_bts_wino", None)
        seq = getattr(w_packed, "_bts_pack_seq", 0)        # bumped by train.WeightPacker whenever it refills this buffer in place
        if uw is not None and getattr(w_packed, "_bts_wino_seq", 0) != seq:
            pack_wino_weight(w_packed, c_in_ld, n_tail, c_out16=getattr(w_packed, "_bts_wino_c16", 0), out=uw)
            w_packed._bts_wino_seq = seq
        if uw is None:
            # which channel tile the library will use for this layer (its own decision on the COMPLETE descriptor, asked
            # once per weight: the two packings differ): 48 -> the 16x16x4 tile
            d.w_wino = w_packed.data_ptr()
            bm_, bn_, kind_ = C.c_int(0), C.c_int(0), C.c_int(0)
            _lib.load_real().bts_conv_plan_f32(C.byref(d), C.byref(bm_), C.byref(bn_), C.byref(kind_))
            # (bn is choose_tile's pure function of c_out, whatever kernel family this particular launch ends up on)
            w_packed._bts_wino_c16 = d.c_out if bn_.value == 48 else 0
            uw = pack_wino_weight(w_packed, c_in_ld, n_tail, c_out16=w_packed._bts_wino_c16)
            w_packed._bts_wino, w_packed._bts_wino_seq = uw, seq
        keep.append(uw)
        d.w_wino = uw.data_ptr()
    cin = c_in_real if c_in_real is not None else c_in_ld
    npix_out = B * H * W
    if n_bundles > 1:                                      # algorithmic FLOPs of the grouped conv are passed in c_in_real
        c_out = c_out * n_bundles                          # (real input channels per OUTPUT channel = channels per group)
    flops = 2.0 * npix_out * c_out * cin * flops_taps      # algorithmic: the reference's 3x3 on the upsampled map
    if algo_flops is not None:
        flops = float(algo_flops)
    nbytes = 4.0 * (B * h_in * w_in * cin + npix_out * c_out + flops_taps * c_out * cin)
    variant = "conv"
    if _trace is not None:
        bm, bn, kind = C.c_int(0), C.c_int(0), C.c_int(0)
        _lib.load().bts_conv_plan_f32(C.byref(d), C.byref(bm), C.byref(bn), C.byref(kind))
        lay = "nchw" if y_nchw is not None else "nhwc"
        if (kind.value & 15) == 6:
            variant = "conv_wino_kernel<%d>" % bn.value
        elif (kind.value & 15) == 5:
            variant = "conv_halo_emu_kernel<%d,k%d>" % (bn.value, 2 if subpixel else 3)
        elif (kind.value & 15) == 4:
            variant = "conv_stem_kernel<%d>" % bn.value
        elif (kind.value & 15) == 3:
            variant = "conv1x1_kernel<%d,%d>" % (bn.value, bm.value // 32)      # <BN, WM>: rows = 32 * WM, as rocprofv3 names it
        elif kind.value & 15:
            variant = "conv_halo_kernel<%d,k%d,%s%s%s%s>" % (bn.value, 2 if subpixel else 3, lay, ",tail" if (kind.value & 15) == 2 else "",
                                                            ",w8" if kind.value & 32 else "", ",dil" if kind.value & 64 else "")
        else:
            variant = "conv_fwd_kernel<%d,%d,%s%s>" % (bm.value, bn.value, lay, ",splitk" if kind.value & 16 else "")
    xflops = 2.0 * npix_out * c_out * (c_in_ld if n_bundles > 1 else cin) * taps
    if _trace is not None and variant.startswith("conv_wino_kernel"):
        # MFMA products the Winograd kernel ISSUES: 16 transform positions x 32 tiles x BN channels x c_in per workgroup
        # (4 instead of 9 per output and input channel, plus the ragged 8x16-pixel tiles and the channels padded to BN)
        bn_w = bn.value
        nwg = B * ((H + 7) // 8) * ((W + 15) // 16) * ((c_out + bn_w - 1) // bn_w)
        xflops = 2.0 * nwg * 16 * 32 * bn_w * c_in_ld
    if _trace is not None:                                 # tap skipping (dilated ASPP branches): FLOPs really issued
        issued, dense = C.c_long(0), C.c_long(0)
        _lib.load().bts_conv_plan_ksteps_f32(C.byref(d), C.byref(issued), C.byref(dense))
        if dense.value > 0:
            xflops *= issued.value / dense.value
    tops = torch_ops()
    if tops is not None:
        geom = [d.x_pix_stride, d.c_in_ld, d.k_pad, d.B, d.h_in, d.w_in, d.up, d.ksize, d.dil, d.stride, d.pad, d.c_out, d.c_out_pad,
                d.pre_relu, d.act, d.y_pix_stride, d.y_nchw, d.subpixel, d.y2_pix_stride, d.res_pix_stride, d.n_bundles, d.precision,
                d.fill_frames]
        pre_s, pre_b = pre if pre is not None else (None, None)
        e1_s, e1_b = e1 if e1 is not None else (None, None)
        e2_s, e2_b = e2 if e2 is not None else (None, None)
        ws3_t = w_packed._bts_split3 if d.w_split else None
        uw_t = w_packed._bts_wino if d.w_wino else None
        run = lambda: _op(lambda: tops.conv_fwd(x2d, w_packed, pre_s, pre_b, e1_s, e1_b, e2_s, e2_b, out, y2_2d, res2d, splitk_ws,
                                                list(tail_planes) if tail_planes else [], ws3_t, uw_t, geom))
    else:
        run = lambda: _lib.load().bts_conv_fwd_f32(C.byref(d), _stream(x2d))
    with torch.cuda.device(x2d.device):
        rc = _launch(variant, tag, flops, nbytes, run, xflops=xflops)
    _lib.check(rc, "bts_conv_fwd_f32")
    return out


def conv_wgrad(x2d: torch.Tensor, B: int, h_in: int, w_in: int, c_in: int, dy2d: torch.Tensor, c_out: int,
               ksize: int, dil: int = 1, stride: int = 1, pad: Optional[int] = None, up: int = 1,
               ws: Optional[torch.Tensor] = None, tag: str = "wgrad", n_bundles: int = 1,
               pre: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, pre_relu: bool = False) -> torch.Tensor:
    """Weight gradient of one bias-free convolution (bts_conv_wgrad_f32): returns dw as [c_out, ksize*ksize, c_in]
    (OHWI).  x2d: [B*h_in*w_in, >=c_in] NHWC view of the forward input, dy2d: [B*H*W, >=c_out] NHWC view of the
    output gradient; c_in and c_out multiples of 4 (pad with zero channels).  ``n_bundles`` > 1: grouped convolution
    as channel bundles (c_in / c_out per bundle): returns the dense blocks [n_bundles, c_out, ksize*ksize, c_in]."""
    xs, xc = _rows2d(x2d, "conv_wgrad")
    ds, dc = _rows2d(dy2d, "conv_wgrad")
    if pad is None:
        pad = dil * (ksize // 2)
    H = (h_in * up + 2 * pad - dil * (ksize - 1) - 1) // stride + 1
    W = (w_in * up + 2 * pad - dil * (ksize - 1) - 1) // stride + 1
    if c_in % 4 or c_out % 4 or c_in * n_bundles > xc or c_out * n_bundles > dc:
        raise BtsHipError("conv_wgrad: c_in/c_out must be multiples of 4 within the views (%d/%d, %d/%d)" % (c_in, xc, c_out, dc))
    if x2d.shape[0] != B * h_in * w_in or dy2d.shape[0] != B * H * W:
        raise BtsHipError("conv_wgrad: views %s / %s do not match B=%d %dx%d -> %dx%d"
                          % (tuple(x2d.shape), tuple(dy2d.shape), B, h_in, w_in, H, W))
    shape = (c_out, ksize * ksize, c_in) if n_bundles <= 1 else (n_bundles, c_out, ksize * ksize, c_in)
    dw = torch.empty(shape, dtype=torch.float32, device=x2d.device)
    d = ConvWgradDesc()
    d.n_bundles = n_bundles if n_bundles > 1 else 0
    if pre is not None:                       # the forward conv saw [relu](x*scale + shift): gather the same thing
        if pre[0].numel() != c_in * max(n_bundles, 1) or pre[1].numel() != pre[0].numel():
            raise BtsHipError("conv_wgrad: pre vectors must have c_in entries")
        d.pre_scale, d.pre_shift, d.pre_relu = pre[0].data_ptr(), pre[1].data_ptr(), int(bool(pre_relu))
    d.x, d.x_pix_stride, d.c_in = x2d.data_ptr(), xs, c_in
    d.dy, d.dy_pix_stride, d.c_out = dy2d.data_ptr(), ds, c_out
    d.B, d.h_in, d.w_in, d.up, d.ksize, d.dil, d.stride, d.pad = B, h_in, w_in, up, ksize, dil, stride, pad
    d.dw = dw.data_ptr()
    if ws is not None:
        _need(ws, "conv_wgrad")
        d.ws, d.ws_floats = ws.data_ptr(), ws.numel()
    taps = ksize * ksize
    flops = 2.0 * B * H * W * c_out * c_in * taps * max(n_bundles, 1)
    nbytes = 4.0 * (B * h_in * w_in * c_in + B * H * W * c_out + taps * c_out * c_in) * max(n_bundles, 1)
    with torch.cuda.device(x2d.device):
        rc = _launch("conv_wgrad_kernel", tag, flops, nbytes, lambda: _lib.load().bts_conv_wgrad_f32(C.byref(d), _stream(x2d)))
    _lib.check(rc, "bts_conv_wgrad_f32")
    return dw


# ------------------------------------------------------------------------------ train-mode BN
def bn_train_ws_floats(npix: int, C: int) -> int:
    return int(_lib.load().bts_bn_train_ws_floats(npix, C))


def bn_train_stats(x2d: torch.Tensor, C: int, gamma, beta, eps: float, momentum: float, running_mean, running_var,
                   ws: torch.Tensor):
    """Batch statistics of NHWC rows (bts_bn_train_stats_f32): returns (mean, invstd, scale, shift), each [C];
    running_mean / running_var (or None) are updated in place."""
    xs, xc = _rows2d(x2d, "bn_train_stats")
    if C % 4 or C > xc:
        raise BtsHipError("bn_train_stats: C must be a multiple of 4 within the view (%d/%d)" % (C, xc))
    out = torch.empty((4, C), dtype=torch.float32, device=x2d.device)
    _need(ws, "bn_train_stats")
    npix = x2d.shape[0]
    with torch.cuda.device(x2d.device):
        rc = _launch("bn_stats_kernel", "bn.stats", 3.0 * npix * C, 4.0 * npix * C,
                     lambda: _lib.load().bts_bn_train_stats_f32(
                         x2d.data_ptr(), xs, npix, C, _ptr(gamma), _ptr(beta), float(eps), float(momentum),
                         _ptr(running_mean), _ptr(running_var), ws.data_ptr(), ws.numel(), out[0].data_ptr(),
                         out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), _stream(x2d)))
    _lib.check(rc, "bts_bn_train_stats_f32")
    return out[0], out[1], out[2], out[3]


def bn_apply(x2d: torch.Tensor, C: int, scale: torch.Tensor, shift: torch.Tensor, relu: bool, y2d: torch.Tensor):
    """y = [relu](x*scale + shift) over NHWC rows (bts_bn_apply_nhwc_f32)."""
    xs, _ = _rows2d(x2d, "bn_apply")
    ys, _ = _rows2d(y2d, "bn_apply")
    npix = x2d.shape[0]
    with torch.cuda.device(x2d.device):
        rc = _launch("bn_apply_kernel", "bn.apply", 2.0 * npix * C, 8.0 * npix * C,
                     lambda: _lib.load().bts_bn_apply_nhwc_f32(x2d.data_ptr(), xs, npix, C, scale.data_ptr(),
                                                               shift.data_ptr(), int(bool(relu)), y2d.data_ptr(), ys,
                                                               _stream(x2d)))
    _lib.check(rc, "bts_bn_apply_nhwc_f32")
    return y2d


def bn_train_backward(x2d: torch.Tensor, dy2d: torch.Tensor, C: int, mean, invstd, scale, shift, relu: bool,
                      ws: torch.Tensor, dx2d: Optional[torch.Tensor]):
    """Backward of batch-statistic BN (+ fused ReLU): returns (dgamma, dbeta); dx2d (or None) is filled in place."""
    xs, _ = _rows2d(x2d, "bn_train_backward")
    ds, _ = _rows2d(dy2d, "bn_train_backward")
    npix = x2d.shape[0]
    if dy2d.shape[0] != npix:
        raise BtsHipError("bn_train_backward: gradient rows %d != input rows %d" % (dy2d.shape[0], npix))
    g = torch.empty((2, C), dtype=torch.float32, device=x2d.device)
    dxs = 0
    if dx2d is not None:
        dxs, _ = _rows2d(dx2d, "bn_train_backward")
    with torch.cuda.device(x2d.device):
        rc = _launch("bn_bwd_kernels", "bn.bwd", 10.0 * npix * C, 20.0 * npix * C,
                     lambda: _lib.load().bts_bn_train_bwd_f32(
                         x2d.data_ptr(), xs, dy2d.data_ptr(), ds, npix, C, mean.data_ptr(), invstd.data_ptr(),
                         scale.data_ptr(), shift.data_ptr(), int(bool(relu)), ws.data_ptr(), ws.numel(),
                         g[0].data_ptr(), g[1].data_ptr(), _ptr(dx2d), dxs, _stream(x2d)))
    _lib.check(rc, "bts_bn_train_bwd_f32")
    return g[0], g[1]


# ------------------------------------------------------------------------------ tail
def pack_planes(planes: Sequence[torch.Tensor], dst2d: torch.Tensor):
    """Interleave 1-channel maps ([B,1,H,W] contiguous each) into dst2d [npix, n] (an NHWC slice)."""
    stride, n = _rows2d(dst2d, "pack_planes")
    if n != len(planes) or not 1 <= n <= 4:
        raise BtsHipError("pack_planes: need 1..4 planes matching the destination slice")
    npix = dst2d.shape[0]
    for p in planes:
        _need(p, "pack_planes")
        if p.numel() != npix or not p.is_contiguous():
            raise BtsHipError("pack_planes: plane size mismatch")
    ptrs = [_ptr(p) for p in planes] + [C.c_void_p(0)] * (4 - n)
    with torch.cuda.device(dst2d.device):
        rc = _lib.load().bts_pack_planes_f32(ptrs[0], ptrs[1], ptrs[2], ptrs[3], n, npix, _ptr(dst2d), stride,
                                             _stream(dst2d))
    _lib.check(rc, "bts_pack_planes_f32")
    return dst2d


def get_depth_forward(iconv1: torch.Tensor, weight: torch.Tensor, max_depth: float,
                      focal: Optional[torch.Tensor], out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """get_depth + scaling (reference bts.py:289-291): iconv1 [B,C,H,W] NCHW -> final_depth [B,1,H,W]."""
    _need(iconv1, "get_depth_forward")
    _need(weight, "get_depth_forward")
    B, Cc, H, W = iconv1.shape
    if not iconv1.is_contiguous() or tuple(weight.shape) != (1, Cc, 3, 3) or not weight.is_contiguous():
        raise BtsHipError("get_depth_forward: need contiguous iconv1 [B,C,H,W] and weight [1,C,3,3]")
    if focal is not None:
        _need(focal, "get_depth_forward")
        if focal.numel() != B or not focal.is_contiguous():
            raise BtsHipError("get_depth_forward: focal must be [B]")
    if out is None:
        out = torch.empty((B, 1, H, W), dtype=torch.float32, device=iconv1.device)
    elif tuple(out.shape) != (B, 1, H, W) or not out.is_contiguous():
        raise BtsHipError("get_depth_forward: out must be contiguous [B,1,H,W]")
    with torch.cuda.device(iconv1.device):
        rc = _launch("get_depth_kernel<%d>" % Cc, "get_depth", 2.0 * 9 * Cc * B * H * W, 4.0 * (Cc + 1) * B * H * W,
                     lambda: _lib.load().bts_get_depth_f32(_ptr(iconv1), _ptr(weight), B, Cc, H, W, float(max_depth),
                                                           _ptr(focal), _ptr(out), _stream(iconv1)))
    _lib.check(rc, "bts_get_depth_f32")
    return out


# --------------------------------------------------------------------------- pooling
def maxpool3x3s2(src2d: torch.Tensor, B: int, h: int, w: int, dst2d: torch.Tensor, dst2_2d: Optional[torch.Tensor] = None):
    ss, Cc = _rows2d(src2d, "maxpool3x3s2")
    ds, dc = _rows2d(dst2d, "maxpool3x3s2")
    ho, wo = (h + 1) // 2, (w + 1) // 2
    if dc != Cc or src2d.shape[0] != B * h * w or dst2d.shape[0] != B * ho * wo:
        raise BtsHipError("maxpool3x3s2: shape mismatch")
    d2p, d2s = C.c_void_p(0), 0
    if dst2_2d is not None:
        d2s, d2c = _rows2d(dst2_2d, "maxpool3x3s2")
        if d2c != Cc or dst2_2d.shape[0] != B * ho * wo:
            raise BtsHipError("maxpool3x3s2: second destination mismatch")
        d2p = _ptr(dst2_2d)
    nbytes = 4.0 * Cc * (B * h * w + B * ho * wo * (2 if dst2_2d is not None else 1))
    with torch.cuda.device(src2d.device):
        rc = _launch("maxpool3x3s2_kernel", "encoder_pool", 0.0, nbytes,
                     lambda: _lib.load().bts_maxpool3x3s2_nhwc_f32(_ptr(src2d), ss, B, h, w, Cc, _ptr(dst2d), ds, d2p, d2s,
                                                                   _stream(src2d)))
    _lib.check(rc, "bts_maxpool3x3s2_nhwc_f32")
    return dst2d


def bn_relu_avgpool2(src2d: torch.Tensor, B: int, h: int, w: int, scale: torch.Tensor, shift: torch.Tensor,
                     dst2d: torch.Tensor):
    ss, Cc = _rows2d(src2d, "bn_relu_avgpool2")
    ds, dc = _rows2d(dst2d, "bn_relu_avgpool2")
    _need(scale, "bn_relu_avgpool2")
    _need(shift, "bn_relu_avgpool2")
    if dc != Cc or src2d.shape[0] != B * h * w or dst2d.shape[0] != B * (h // 2) * (w // 2) or scale.numel() != Cc:
        raise BtsHipError("bn_relu_avgpool2: shape mismatch")
    nbytes = 4.0 * Cc * (B * h * w + B * (h // 2) * (w // 2))
    with torch.cuda.device(src2d.device):
        rc = _launch("bn_relu_avgpool2_kernel", "encoder_pool", 0.0, nbytes,
                     lambda: _lib.load().bts_bn_relu_avgpool2_nhwc_f32(_ptr(src2d), ss, B, h, w, Cc, _ptr(scale), _ptr(shift),
                                                                       _ptr(dst2d), ds, _stream(src2d)))
    _lib.check(rc, "bts_bn_relu_avgpool2_nhwc_f32")
    return dst2d
