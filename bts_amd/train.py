"""Training-step graph of the BTS decoder (reference: bts.forward in train() mode,
pytorch/bts.py:223-293, driven by bts_main.py:476-500).

Every convolution of the path -- forward, input gradient and weight gradient -- runs on the hand-written gfx950
kernels of libbts_hip.so:

* forward      : bts_conv_fwd_f32 (the inference kernel, plain epilogue),
* input grad   : the same kernel on the output gradient with flipped/transposed weights
                 (a stride-1 convolution's adjoint is a convolution: pad' = dil*(k-1) - pad),
* weight grad  : bts_conv_wgrad_f32 (K = pixels GEMM with a deterministic split over pixels),
* LPG          : bts_lpg_fwd_f32 / bts_lpg_bwd_f32 through ops.LpgFunction.

What is cheap and memory-bound (batch-statistic BN, ELU/ReLU/sigmoid, sin/cos of the plane parameters, concat,
normalize) stays on PyTorch-ROCm's precompiled elementwise/reduction kernels, kept in channels_last so the HIP
convolutions read and write them in place as NHWC.  MIOpen is bypassed (no gfx950 find-db in this image).

There is no CPU path here: non-CUDA tensors raise.
"""
from __future__ import annotations

import weakref
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

from . import ops
from ._lib import BtsHipError

FUSED_DENSE_BLOCKS = True          # DenseNet blocks as one in-place autograd node (False: layer-by-layer graph with torch.cat)
_WS: Dict[Tuple[str, int], torch.Tensor] = {}
WGRAD_WS_FLOATS = 48 << 20        # 192 MB of split-K partials per (device, stream)


def _workspace(device: torch.device) -> torch.Tensor:
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    ws = _WS.get(key)
    if ws is None:
        ws = _WS[key] = torch.empty(WGRAD_WS_FLOATS, dtype=torch.float32, device=device)
    return ws


_SPLITK_WS: Dict[Tuple[str, int], torch.Tensor] = {}
SPLITK_WS_FLOATS = 16 << 20       # 64 MB: lets the forward / input-gradient launches of the deep, pixel-starved layers
                                  # split K (bts_conv_desc.splitk_ws); larger layers never split


def _splitk_workspace(device: torch.device) -> torch.Tensor:
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    ws = _SPLITK_WS.get(key)
    if ws is None:
        ws = _SPLITK_WS[key] = torch.empty(SPLITK_WS_FLOATS, dtype=torch.float32, device=device)
    return ws


def _nhwc_rows(x: torch.Tensor) -> Tuple[torch.Tensor, int]:
    """[B,C,H,W] (any strides) -> ([B*H*W, C4] NHWC rows with C padded to a multiple of 4, C4).
    A channels_last tensor with C % 4 == 0 is viewed, not copied -- and so is a CHANNEL SLICE of one (what autograd hands
    back for the inputs of a torch.cat, and what a dense block's layers read): the kernels take a pixel stride."""
    B, C, H, W = x.shape
    c4 = ops.round_up(C, 4)
    if c4 == C and B * H * W > 0 and x.stride(1) == 1 and W > 1 and H > 1:
        ct = x.stride(3)
        if (ct >= C and ct % 4 == 0 and x.stride(2) == W * ct and (B == 1 or x.stride(0) == H * W * ct)
                and x.data_ptr() % 16 == 0):
            return x.as_strided((B * H * W, C), (ct, 1)), c4
    rows = x.permute(0, 2, 3, 1)
    if c4 != C:
        rows = F.pad(rows, (0, c4 - C))
    return rows.contiguous().view(B * H * W, c4), c4


class WeightPacker:
    """Packed copies of the convolution weights for the training step.

    The optimiser rewrites the OIHW parameters every iteration; the HIP kernels want them as [c_out_pad][k_pad]
    (forward) and as the same layout of the flipped, transposed kernel (input gradient).  Entries are registered on
    first use; ``refresh()`` -- called at the top of each training forward -- re-packs every entry whose parameter
    changed (``Tensor._version``) in ONE launch of bts_pack_weights_f32.  A stale entry met later is re-packed on the
    spot, so correctness never depends on ``refresh`` having been called."""

    FWD, DGRAD = 0, 1

    def __init__(self):
        self.entries = {}          # (data_ptr, shape, c_in_ld, mode, groups) -> dict(wref, dst, version, rows)
        self._table = None
        self._table_key = None

    @staticmethod
    def bundle_channels(weight, groups):
        """Channels per bundle for a grouped weight [C, C/groups, k, k]: consecutive groups packed block-diagonally
        into bundles of max(32, C/groups) channels (ops.pack_grouped_conv_weight uses the same rule)."""
        c, cg = weight.shape[0], weight.shape[1]
        cb = max(32, cg)
        if groups * cg != c or cb % cg or c % cb or cb % 32:
            raise BtsHipError("train.conv2d: grouped convolution %s with %d groups is not built" % (tuple(weight.shape), groups))
        return cb

    @staticmethod
    def _rows(weight, c_in_ld, mode, groups):
        """Table rows (without pointers) + destination shape for one weight: one row for a dense convolution, one per
        channel bundle for a grouped one.  Row = (src offset, dst offset, rows, inner, k, c_in_ld, rows_pad, k_pad,
        s_row, s_c, flip, cg, gmode)."""
        cout, cin_g, k, _ = weight.shape
        kk = k * k
        if groups == 1:
            if mode == WeightPacker.FWD:
                rows, inner, s_row, s_c, flip = cout, cin_g, cin_g * kk, kk, 0
            else:
                rows, inner, s_row, s_c, flip = cin_g, cout, kk, cin_g * kk, 1
            rows_pad, k_pad = ops.round_up(rows, 32), ops.round_up(kk * c_in_ld, 32)
            return [(0, 0, rows, inner, k, c_in_ld, rows_pad, k_pad, s_row, s_c, flip, 0, 0)], (rows_pad, k_pad)
        cb = WeightPacker.bundle_channels(weight, groups)
        if c_in_ld != cb:
            raise BtsHipError("train.conv2d: grouped convolutions are packed with c_in_ld = bundle width")
        nb, k_pad = cout // cb, kk * cb
        if mode == WeightPacker.FWD:
            s_row, s_c, flip, gmode = cin_g * kk, kk, 0, 1
        else:
            s_row, s_c, flip, gmode = kk, cin_g * kk, 1, 2
        rows = [(j * cb * cin_g * kk, j * cb * k_pad, cb, cb, k, cb, cb, k_pad, s_row, s_c, flip, cin_g, gmode)
                for j in range(nb)]
        return rows, (nb, cb, k_pad)

    def _table_for(self, items):
        import numpy as np
        from . import _lib
        lib = _lib.load()
        table, first = [], 0
        for e in items:
            src, dst = e["ptr"], e["dst"].data_ptr()
            for (so, do, rows, inner, k, cld, rows_pad, k_pad, s_row, s_c, flip, cg, gmode) in e["rows"]:
                table.append([src + 4 * so, dst + 4 * do, rows, inner, k, cld, rows_pad, k_pad, s_row, s_c, flip, first,
                              cg, gmode])
                first += int(lib.bts_pack_weights_blocks(rows_pad, k_pad))
        return torch.from_numpy(np.asarray(table, dtype=np.int64)).to(items[0]["dst"].device), first, len(table)

    def _launch(self, items, table, blocks, n_rows):
        from . import _lib
        dev = items[0]["dst"].device
        with torch.cuda.device(dev):
            rc = ops._launch("pack_weights_kernel", "pack", 0.0, 0.0,
                             lambda: _lib.load().bts_pack_weights_f32(table.data_ptr(), n_rows, blocks,
                                                                      torch.cuda.current_stream(dev).cuda_stream))
        _lib.check(rc, "bts_pack_weights_f32")
        for e in items:
            e["version"] = e["wref"]()._version
            # derived forms cached on the packed buffer (Winograd U, bf16 planes: ops.conv_forward) are now stale
            e["dst"]._bts_pack_seq = getattr(e["dst"], "_bts_pack_seq", 0) + 1

    def get(self, weight, c_in_ld, mode, groups=1):
        """Packed buffer for ``weight`` (an nn.Parameter or any contiguous OIHW CUDA tensor), current with its values:
        [rows_pad, k_pad] for a dense convolution, [n_bundles, cb, k_pad] for a grouped one."""
        key = (weight.data_ptr(), tuple(weight.shape), c_in_ld, mode, groups)
        e = self.entries.get(key)
        if e is not None and e["wref"]() is None:          # the tensor this entry was packed from is gone: the address
            e = None                                        # now belongs to someone else's values
        if e is None:
            if not weight.is_contiguous():
                raise BtsHipError("train.conv2d: weight must be contiguous OIHW")
            rows, shape = self._rows(weight, c_in_ld, mode, groups)
            e = self.entries[key] = dict(wref=weakref.ref(weight), ptr=weight.data_ptr(), rows=rows, version=-1,
                                         dst=torch.empty(shape, dtype=torch.float32, device=weight.device))
            self._table_key = None
        if e["version"] != e["wref"]()._version:
            table, blocks, n_rows = self._table_for([e])
            self._launch([e], table, blocks, n_rows)
        return e["dst"]

    def refresh(self, force: bool = False):
        """Re-pack every registered weight whose parameter changed since its last packing -- with ``force``: every
        registered weight -- in one launch."""
        dead = [k for k, e in self.entries.items() if e["wref"]() is None]
        for k in dead:
            del self.entries[k]
        stale = [e for e in self.entries.values() if force or e["version"] != e["wref"]()._version]
        if not stale:
            return
        key = tuple(id(e) for e in stale)
        if self._table_key != key:                        # the usual case after step 1: the same full set every step
            self._table, self._table_blocks, self._table_rows = self._table_for(stale)
            self._table_key = key
        self._launch(stale, self._table, self._table_blocks, self._table_rows)


_PACKER = WeightPacker()


def begin_step():
    """Top of a training forward: bring all packed weights up to date with one launch.

    Every registered weight is re-packed, changed or not: ``Tensor._version`` is NOT a reliable change signal in a
    training loop -- torch's fused optimisers (``AdamW(fused=True)``, ``_fused_adamw_``) rewrite the parameters without
    bumping it -- and in steady state the optimiser has touched every weight anyway (one 0.5 ms launch per step).  For
    the same reason a training forward also ages the eval-mode packs and recorded graphs/plans of every model
    (``workspace.invalidate_packs``): the next eval forward re-packs from the current values."""
    if _STEP_DEPTH[0] > 0:                 # inside model_step(): the model's forward already did this
        return
    from . import workspace
    workspace.invalidate_packs()
    _PACKER.refresh(force=True)


_STEP_DEPTH = [0]


class model_step:
    """``with train.model_step():`` around a whole-model training forward: one begin_step() for encoder + decoder."""

    def __enter__(self):
        begin_step()
        _STEP_DEPTH[0] += 1

    def __exit__(self, *exc):
        _STEP_DEPTH[0] -= 1
        return False


class _ConvFn(torch.autograd.Function):
    """y = conv2d(nearest_up(x, up), w, stride, padding, dilation, groups), bias-free, on libbts_hip.so."""

    @staticmethod
    def forward(ctx, x, weight, stride, padding, dilation, up, tag, groups, act=ops.ACT_NONE):
        ops._need(x, "train.conv2d")
        ops._need(weight, "train.conv2d")
        B, C, h, w = x.shape
        cout, cin_g, k, k2 = weight.shape
        if cin_g * groups != C or k != k2:
            raise BtsHipError("train.conv2d: weight %s (groups %d) does not fit input %s"
                              % (tuple(weight.shape), groups, tuple(x.shape)))
        x2d, c4 = _nhwc_rows(x.detach())
        H = (h * up + 2 * padding - dilation * (k - 1) - 1) // stride + 1
        W = (w * up + 2 * padding - dilation * (k - 1) - 1) // stride + 1
        y = torch.empty((B, H, W, cout), dtype=torch.float32, device=x.device)
        if act not in (ops.ACT_NONE, ops.ACT_ELU):
            raise BtsHipError("train.conv2d: only ELU can ride in the epilogue (its derivative is a function of y)")
        if groups > 1:
            if up != 1 or cout != C or act != ops.ACT_NONE:
                raise BtsHipError("train.conv2d: grouped convolutions need cin == cout, no upsample, no activation")
            cb = WeightPacker.bundle_channels(weight, groups)
            wp = _PACKER.get(weight, cb, WeightPacker.FWD, groups)
            ops.conv_forward(x2d, B, h, w, wp, cb, k, dil=dilation, c_in_ld=cb, y2d=y.view(B * H * W, cout), stride=stride,
                             pad=padding, tag=tag + ".fwd", c_in_real=cin_g, n_bundles=C // cb)
        else:
            wp = _PACKER.get(weight, c4, WeightPacker.FWD)
            ops.conv_forward(x2d, B, h, w, wp, cout, k, dil=dilation, up=up, c_in_ld=c4, y2d=y.view(B * H * W, cout),
                             stride=stride, pad=padding, tag=tag + ".fwd", c_in_real=C, act=act,
                             splitk_ws=_splitk_workspace(x.device))
        out = y.permute(0, 3, 1, 2)           # [B,cout,H,W] channels_last view
        if act == ops.ACT_ELU:
            ctx.save_for_backward(x2d, out)   # ELU'(v) = 1 (y > 0) or y + 1: the pre-activation is never stored
        else:
            ctx.save_for_backward(x2d)
        ctx.weight = weight                   # the caller's parameter object: the packer tracks it by identity/version
        ctx.geom = (B, C, h, w, c4, cout, k, stride, padding, dilation, up, H, W, tag, groups)
        ctx.act = act
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x2d = ctx.saved_tensors[0]
        weight = ctx.weight
        B, C, h, w, c4, cout, k, stride, padding, dilation, up, H, W, tag, groups = ctx.geom
        if ctx.act == ops.ACT_ELU:
            grad_out = torch.ops.aten.elu_backward(grad_out, 1.0, 1.0, 1.0, True, ctx.saved_tensors[1])
        dy2d, co4 = _nhwc_rows(grad_out)
        dev = grad_out.device
        dx = dw = None
        if ctx.needs_input_grad[0]:
            pad_t = dilation * (k - 1) - padding
            if stride not in (1, 2) or pad_t < 0 or (stride == 2 and up != 1):
                raise BtsHipError("train.conv2d: input gradient is built for stride 1 or 2 with padding <= "
                                  "dilation*(k-1); got stride %d padding %d" % (stride, padding))
            Hs, Ws = h * up, w * up
            g2d, gh, gw = dy2d, H, W
            if stride == 2:
                # adjoint of the stride: the gradient sits on the even positions of a zero map of the input's size, and
                # the stride-1 adjoint below runs on that (4x the FLOPs of the handful of strided layers, exact)
                u = torch.zeros((B, Hs, Ws, co4), dtype=torch.float32, device=dev)
                u[:, 0:2 * H:2, 0:2 * W:2] = (dy2d.view(B, H, W, co4) if dy2d.is_contiguous()
                                              else grad_out.permute(0, 2, 3, 1))      # a strided view has co4 == cout
                g2d, gh, gw = u.view(B * Hs * Ws, co4), Hs, Ws
            dxu = torch.empty((B, Hs, Ws, C), dtype=torch.float32, device=dev)
            if groups > 1:
                cb = WeightPacker.bundle_channels(weight, groups)
                wp = _PACKER.get(weight, cb, WeightPacker.DGRAD, groups)
                ops.conv_forward(g2d, B, gh, gw, wp, cb, k, dil=dilation, c_in_ld=cb, y2d=dxu.view(B * Hs * Ws, C),
                                 pad=pad_t, tag=tag + ".dgrad", c_in_real=weight.shape[1], n_bundles=C // cb)
            else:
                wp = _PACKER.get(weight, co4, WeightPacker.DGRAD)        # flipped, transposed kernel [cin][taps*co4]
                ops.conv_forward(g2d, B, gh, gw, wp, C, k, dil=dilation, c_in_ld=co4, y2d=dxu.view(B * Hs * Ws, C),
                                 pad=pad_t, tag=tag + ".dgrad", c_in_real=cout, splitk_ws=_splitk_workspace(dev))
            if up == 2:                                                 # adjoint of the nearest-2x upsample
                dxu = dxu.view(B, h, 2, w, 2, C).sum(dim=(2, 4))
            dx = dxu.permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            if groups > 1:
                cb = WeightPacker.bundle_channels(weight, groups)
                cg, nb = weight.shape[1], C // cb
                gpb = cb // cg
                g = ops.conv_wgrad(x2d, B, h, w, cb, dy2d, cb, k, dil=dilation, stride=stride, pad=padding,
                                   ws=_workspace(dev), tag=tag + ".wgrad", n_bundles=nb)   # [nb, cb, taps, cb] dense blocks
                g6 = g.view(nb, gpb, cg, k * k, gpb, cg)
                idx = torch.arange(gpb, device=dev)
                diag = g6[:, idx, :, :, idx, :]                         # [gpb, nb, cg_out, taps, cg_in]: each group's own block
                dw = diag.permute(1, 0, 2, 4, 3).reshape(cout, cg, k, k).contiguous()
            else:
                g = ops.conv_wgrad(x2d, B, h, w, c4, dy2d, co4, k, dil=dilation, stride=stride, pad=padding, up=up,
                                   ws=_workspace(dev), tag=tag + ".wgrad")
                dw = g[:cout, :, :C].reshape(cout, k, k, C).permute(0, 3, 1, 2).contiguous()   # OIHW, the layout DDP buckets expect
        return dx, dw, None, None, None, None, None, None, None


def conv2d(x: torch.Tensor, weight: torch.Tensor, padding: int = 0, dilation: int = 1, stride: int = 1,
           up: int = 1, tag: str = "conv", groups: int = 1, act: int = ops.ACT_NONE) -> torch.Tensor:
    """Differentiable bias-free convolution on the HIP kernels; ``up=2`` folds a nearest-2x upsample of ``x`` into
    the gather (reference upconv, bts.py:90-92); ``groups`` > 1: ResNeXt's grouped 3x3 as channel bundles;
    ``act=ops.ACT_ELU``: the ELU that follows every decoder convolution (bts.py:93, 183-221) applied in the epilogue.
    Returns a channels_last [B,c_out,H,W] tensor."""
    return _ConvFn.apply(x, weight, stride, padding, dilation, up, tag, groups, act)


_BN_WS: Dict[Tuple[str, int], torch.Tensor] = {}


def _bn_workspace(device: torch.device, floats: int) -> torch.Tensor:
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    ws = _BN_WS.get(key)
    if ws is None or ws.numel() < floats:
        ws = _BN_WS[key] = torch.empty(max(floats, 1 << 20), dtype=torch.float32, device=device)
    return ws


_PENDING_COUNTS: Dict[int, torch.Tensor] = {}
_DEFER_DEPTH = [0]


def _count_batch(bn):
    """``num_batches_tracked += 1`` (nn.BatchNorm2d.forward in train() mode).  Inside an encoder / decoder forward
    (``_deferred_batch_counts``) the ~175 one-element kernels of a step are collected into one multi-tensor add issued
    when the forward returns; a norm layer run on its own counts immediately."""
    t = bn.num_batches_tracked
    if _DEFER_DEPTH[0] == 0:
        t.add_(1)
        return
    if id(t) in _PENDING_COUNTS:          # the same layer twice in one forward: settle the first count now
        flush_batch_counts()
    _PENDING_COUNTS[id(t)] = t


def flush_batch_counts():
    if _PENDING_COUNTS:
        ts = list(_PENDING_COUNTS.values())
        _PENDING_COUNTS.clear()
        with torch.no_grad():
            torch._foreach_add_(ts, 1)


class _deferred_batch_counts:
    def __enter__(self):
        _DEFER_DEPTH[0] += 1

    def __exit__(self, *exc):
        _DEFER_DEPTH[0] -= 1
        if _DEFER_DEPTH[0] == 0:
            flush_batch_counts()
        return False


class _BnFn(torch.autograd.Function):
    """y = [relu](batch_norm(x)) with batch statistics, on libbts_hip.so (bn_train.hip); the module's running
    buffers are updated in place exactly as nn.BatchNorm2d.train() does."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, relu):
        B, C, H, W = x.shape
        x2d, _ = _nhwc_rows(x.detach())
        npix = B * H * W
        ws = _bn_workspace(x.device, ops.bn_train_ws_floats(npix, C))
        track = bn.track_running_stats and bn.running_mean is not None
        mean, invstd, scale, shift = ops.bn_train_stats(
            x2d, C, None if gamma is None else gamma.detach(), None if beta is None else beta.detach(), bn.eps,
            bn.momentum, bn.running_mean if track else None, bn.running_var if track else None, ws)
        if track and bn.num_batches_tracked is not None:
            _count_batch(bn)
        y = torch.empty((B, H, W, C), dtype=torch.float32, device=x.device)
        ops.bn_apply(x2d, C, scale, shift, relu, y.view(npix, C))
        ctx.save_for_backward(x2d, mean, invstd, scale, shift)
        ctx.meta = (B, C, H, W, relu, gamma is not None)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, grad_out):
        x2d, mean, invstd, scale, shift = ctx.saved_tensors
        B, C, H, W, relu, affine = ctx.meta
        dy2d, _ = _nhwc_rows(grad_out)
        npix = B * H * W
        dx = torch.empty((B, H, W, C), dtype=torch.float32, device=grad_out.device) if ctx.needs_input_grad[0] else None
        ws = _bn_workspace(grad_out.device, ops.bn_train_ws_floats(npix, C))
        dgamma, dbeta = ops.bn_train_backward(x2d, dy2d, C, mean, invstd, scale, shift, relu, ws,
                                              None if dx is None else dx.view(npix, C))
        return (None if dx is None else dx.permute(0, 3, 1, 2), dgamma if affine and ctx.needs_input_grad[1] else None,
                dbeta if affine and ctx.needs_input_grad[2] else None, None, None)


def _bn(x: torch.Tensor, bn: torch.nn.BatchNorm2d, relu: bool = False) -> torch.Tensor:
    """nn.BatchNorm2d (+ the ReLU that follows it) inside the training graph.  train() mode with a fixed momentum and
    C % 4 == 0 runs on the HIP kernels; a frozen (eval-mode) layer or an exotic configuration runs on ATen's native
    kernels (MIOpen bypassed: no gfx950 find-db in this image)."""
    if bn.training and bn.momentum is not None and x.shape[1] % 4 == 0 and x.is_cuda and x.dtype == torch.float32:
        return _BnFn.apply(x, bn.weight, bn.bias, bn, relu)
    with torch.backends.cudnn.flags(enabled=False):
        y = bn(x)
    return F.relu(y) if relu else y


# ------------------------------------------------------------------------------------------ module forwards
def atrous_forward(m, x):
    """atrous_conv.forward, bts.py:79-80."""
    seq = m.atrous_conv.aconv_sequence
    x = _bn(x, m.atrous_conv.first_bn, relu=True) if m.apply_bn_first else F.relu(x)
    x = _bn(conv2d(x, seq[1].weight, tag="aspp1x1"), seq[2], relu=True)
    return conv2d(x, seq[4].weight, padding=m.dilation, dilation=m.dilation, tag="aspp3x3")


def upconv_forward(m, x):
    """upconv.forward, bts.py:90-94."""
    if m.ratio not in (1, 2):
        raise BtsHipError("upconv: ratio %r not built (1 or 2)" % (m.ratio,))
    return conv2d(x, m.conv.weight, padding=1, up=int(m.ratio), tag="upconv", act=ops.ACT_ELU)


def reduction_forward(m, net):
    """reduction_1x1.forward, bts.py:124-136: the 1x1 stack, then (non-final) the plane-parameter transform."""
    for layer in m.reduc:
        if isinstance(layer, torch.nn.Conv2d):                       # plane_params
            net = conv2d(net, layer.weight, tag="reduc")
        elif isinstance(layer[1], torch.nn.ELU) and layer[1].alpha == 1.0:
            net = conv2d(net, layer[0].weight, tag="reduc", act=ops.ACT_ELU)   # inter_*: ELU in the epilogue
        else:
            net = layer[1](conv2d(net, layer[0].weight, tag="reduc"))   # Sigmoid (final)
    if m.is_final:
        return net
    import math
    theta = torch.sigmoid(net[:, 0]) * (math.pi / 3)
    phi = torch.sigmoid(net[:, 1]) * (math.pi * 2)
    dist = torch.sigmoid(net[:, 2]) * m.max_depth
    sin_t = torch.sin(theta)
    return torch.stack([sin_t * torch.cos(phi), sin_t * torch.sin(phi), torch.cos(theta), dist], dim=1)


def _conv_elu(seq, x, tag):
    return conv2d(x, seq[0].weight, padding=1, tag=tag, act=ops.ACT_ELU)


def decoder_forward(dec, features, focal):
    """bts.forward in train() mode (bts.py:223-293): same dataflow as the inference path, as an autograd graph."""
    with _deferred_batch_counts():
        return _decoder_forward(dec, features, focal)


def _decoder_forward(dec, features, focal):
    begin_step()
    skip0, skip1, skip2, skip3 = features[1], features[2], features[3], features[4]
    md = dec.params.max_depth
    cl = torch.channels_last
    x = F.relu(features[5]).contiguous(memory_format=cl)
    x = _bn(upconv_forward(dec.upconv5, x), dec.bn5)
    iconv5 = _conv_elu(dec.conv5, torch.cat([x, skip3], 1), "conv5")
    x = _bn(upconv_forward(dec.upconv4, iconv5), dec.bn4)
    concat4 = torch.cat([x, skip2], 1)
    iconv4 = _bn(_conv_elu(dec.conv4, concat4, "conv4"), dec.bn4_2)

    grown, branches, inp = concat4, [], iconv4
    for name in ("daspp_3", "daspp_6", "daspp_12", "daspp_18", "daspp_24"):
        d = atrous_forward(getattr(dec, name), inp)
        branches.append(d)
        grown = torch.cat([grown, d], 1)
        inp = grown
    daspp_feat = _conv_elu(dec.daspp_conv, torch.cat([iconv4] + branches, 1), "daspp_conv")

    def lpg_scale(reduc_mod, lpg_mod, feat):
        r = reduction_forward(reduc_mod, feat)
        plane_eq = torch.cat([F.normalize(r[:, :3], 2, 1), r[:, 3:4]], 1).contiguous()
        return lpg_mod(plane_eq, focal).unsqueeze(1) / md

    depth_8x8 = lpg_scale(dec.reduc8x8, dec.lpg8x8, daspp_feat)
    x = _bn(upconv_forward(dec.upconv3, daspp_feat), dec.bn3)
    iconv3 = _conv_elu(dec.conv3, torch.cat([x, skip1, depth_8x8[:, :, ::4, ::4]], 1), "conv3")
    depth_4x4 = lpg_scale(dec.reduc4x4, dec.lpg4x4, iconv3)
    x = _bn(upconv_forward(dec.upconv2, iconv3), dec.bn2)
    iconv2 = _conv_elu(dec.conv2, torch.cat([x, skip0, depth_4x4[:, :, ::2, ::2]], 1), "conv2")
    depth_2x2 = lpg_scale(dec.reduc2x2, dec.lpg2x2, iconv2)
    upconv1 = upconv_forward(dec.upconv1, iconv2)
    reduc1x1 = reduction_forward(dec.reduc1x1, upconv1)
    iconv1 = _conv_elu(dec.conv1, torch.cat([upconv1, reduc1x1, depth_2x2, depth_4x4, depth_8x8], 1), "conv1")
    final_depth = md * torch.sigmoid(conv2d(iconv1, dec.get_depth[0].weight, padding=1, tag="get_depth"))
    if dec.params.dataset == 'kitti':
        final_depth = final_depth * focal.view(-1, 1, 1, 1).float() / 715.0873
    return depth_8x8, depth_4x4, depth_2x2, reduc1x1, final_depth, iconv1


# ------------------------------------------------------------------------------------------ encoder (DenseNet)
class _DenseBlockFn(torch.autograd.Function):
    """One torchvision _DenseBlock in train() mode as a single autograd node working IN PLACE on one NHWC buffer.

    torchvision re-concatenates the whole prefix for every layer (and autograd then splits and re-adds the prefix
    gradient once per layer: O(L^2) copy/add launches, 36 layers in DenseNet161's third block).  Here the block owns one
    [B,H,W,C_total] buffer: layer i normalises the first C_i channels (a strided view), and its 3x3 convolution writes
    its growth channels straight into columns [C_i, C_i + g).  Backward walks the layers in reverse on one gradient
    buffer of the same shape: each layer reads its own columns as dy and adds its input gradient onto the prefix.
    Saved per layer: the 1x1 output and the four BN vectors of both norms; the normalised inputs are never materialised --
    norm + ReLU ride in the prologue of the forward convolutions and of the weight-gradient gather."""

    @staticmethod
    def forward(ctx, x, block, *params):
        B, C0, H, W = x.shape
        layers = list(block.values())
        g = layers[0].conv2.out_channels
        mid = layers[0].conv1.out_channels
        Ct = C0 + len(layers) * g
        npix = B * H * W
        dev = x.device
        buf = torch.empty((npix, Ct), dtype=torch.float32, device=dev)
        buf[:, :C0] = x.detach().permute(0, 2, 3, 1).reshape(npix, C0)
        saved = []
        ws = _bn_workspace(dev, ops.bn_train_ws_floats(npix, Ct))
        sk = _splitk_workspace(dev)
        for i, L in enumerate(layers):
            Ci = C0 + i * g
            s1 = ops.bn_train_stats(buf[:, :Ci], Ci, L.norm1.weight.detach(), L.norm1.bias.detach(), L.norm1.eps,
                                    L.norm1.momentum, L.norm1.running_mean, L.norm1.running_var, ws)
            # norm + ReLU ride in the convolutions' prologue (applied on the way to LDS): the normalised tensors are
            # never written in the forward pass
            t1 = torch.empty((npix, mid), dtype=torch.float32, device=dev)
            ops.conv_forward(buf[:, :Ci], B, H, W, _PACKER.get(L.conv1.weight, Ci, WeightPacker.FWD), mid, 1, c_in_ld=Ci,
                             pre=(s1[2], s1[3]), pre_relu=True, y2d=t1, tag="enc.fwd", splitk_ws=sk)
            s2 = ops.bn_train_stats(t1, mid, L.norm2.weight.detach(), L.norm2.bias.detach(), L.norm2.eps,
                                    L.norm2.momentum, L.norm2.running_mean, L.norm2.running_var, ws)
            ops.conv_forward(t1, B, H, W, _PACKER.get(L.conv2.weight, mid, WeightPacker.FWD), g, 3, c_in_ld=mid,
                             pre=(s2[2], s2[3]), pre_relu=True, y2d=buf[:, Ci:Ci + g], tag="enc.fwd", splitk_ws=sk)
            for bn in (L.norm1, L.norm2):
                if bn.num_batches_tracked is not None:
                    _count_batch(bn)
            saved += [t1, torch.stack(s1), torch.stack(s2)]
        ctx.save_for_backward(buf, *saved)
        ctx.block = block
        ctx.geom = (B, C0, H, W, g, mid, Ct)
        return buf.view(B, H, W, Ct).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, grad_out):
        buf, *saved = ctx.saved_tensors
        B, C0, H, W, g, mid, Ct = ctx.geom
        layers = list(ctx.block.values())
        npix = B * H * W
        dev = grad_out.device
        G = torch.empty((npix, Ct), dtype=torch.float32, device=dev)     # private copy: the walk below accumulates into it
        G.view(B, H, W, Ct).copy_(grad_out.permute(0, 2, 3, 1))
        d_a2 = torch.empty((npix, mid), dtype=torch.float32, device=dev)
        d_t1 = torch.empty((npix, mid), dtype=torch.float32, device=dev)
        d_a1 = torch.empty((npix, Ct), dtype=torch.float32, device=dev)
        dpre = torch.empty((npix, Ct), dtype=torch.float32, device=dev)
        ws = _bn_workspace(dev, ops.bn_train_ws_floats(npix, Ct))
        wws = _workspace(dev)
        sk = _splitk_workspace(dev)
        grads = [None] * (6 * len(layers))
        need = ctx.needs_input_grad
        for i in range(len(layers) - 1, -1, -1):
            L = layers[i]
            Ci = C0 + i * g
            t1, s1, s2 = saved[3 * i], saved[3 * i + 1], saved[3 * i + 2]
            gy = G[:, Ci:Ci + g]                                               # this layer's output gradient, in place
            ops.conv_forward(gy, B, H, W, _PACKER.get(L.conv2.weight, g, WeightPacker.DGRAD), mid, 3, c_in_ld=g, y2d=d_a2,
                             pad=1, tag="enc.dgrad", splitk_ws=sk)
            if need[2 + 6 * i + 5]:                # the weight-gradient gather re-applies norm2 + ReLU to t1 on the fly
                w2 = ops.conv_wgrad(t1, B, H, W, mid, gy, g, 3, ws=wws, tag="enc.wgrad", pre=(s2[2], s2[3]), pre_relu=True)
                grads[6 * i + 5] = w2.reshape(g, 3, 3, mid).permute(0, 3, 1, 2).contiguous()
            dg2, db2 = ops.bn_train_backward(t1, d_a2, mid, s2[0], s2[1], s2[2], s2[3], True, ws, d_t1)
            if need[2 + 6 * i + 3]:
                grads[6 * i + 3] = dg2
            if need[2 + 6 * i + 4]:
                grads[6 * i + 4] = db2
            ops.conv_forward(d_t1, B, H, W, _PACKER.get(L.conv1.weight, mid, WeightPacker.DGRAD), Ci, 1, c_in_ld=mid,
                             y2d=d_a1[:, :Ci], pad=0, tag="enc.dgrad", splitk_ws=sk)
            if need[2 + 6 * i + 2]:
                w1 = ops.conv_wgrad(buf[:, :Ci], B, H, W, Ci, d_t1, mid, 1, ws=wws, tag="enc.wgrad", pre=(s1[2], s1[3]),
                                    pre_relu=True)
                grads[6 * i + 2] = w1.reshape(mid, Ci, 1, 1)
            dg1, db1 = ops.bn_train_backward(buf[:, :Ci], d_a1[:, :Ci], Ci, s1[0], s1[1], s1[2], s1[3], True, ws, dpre[:, :Ci])
            if need[2 + 6 * i + 0]:
                grads[6 * i + 0] = dg1
            if need[2 + 6 * i + 1]:
                grads[6 * i + 1] = db1
            G[:, :Ci] += dpre[:, :Ci]                                          # the prefix receives this layer's input gradient
        dx = G[:, :C0].reshape(B, H, W, C0).permute(0, 3, 1, 2) if need[0] else None
        return (dx, None) + tuple(grads)


def _dense_block(block, x):
    """Fused in-place dense block when every norm layer is an ordinary train()-mode BatchNorm2d with affine parameters
    and channel counts the kernels take (multiples of 4); otherwise the generic layer-by-layer graph."""
    nn = torch.nn
    layers = list(block.values())
    ok = x.is_cuda and x.dtype == torch.float32 and x.shape[1] % 4 == 0
    for L in layers:
        for bn in (L.norm1, L.norm2):
            ok = ok and isinstance(bn, nn.BatchNorm2d) and bn.training and bn.momentum is not None and bn.affine \
                and bn.track_running_stats and bn.running_mean is not None
        ok = ok and L.conv1.bias is None and L.conv2.bias is None and L.conv2.out_channels % 4 == 0 \
            and L.conv1.out_channels % 4 == 0 and L.conv2.padding[0] == 1 and L.conv1.kernel_size[0] == 1 \
            and L.conv2.kernel_size[0] == 3
    if not ok:
        return None
    params = []
    for L in layers:
        params += [L.norm1.weight, L.norm1.bias, L.conv1.weight, L.norm2.weight, L.norm2.bias, L.conv2.weight]
    return _DenseBlockFn.apply(x, block, *params)


def _run_children(children, x, tapped=None, taps=None):
    """Run (name, module) pairs in order on the HIP kernels: bias-free ungrouped convolutions, batch-statistic BN
    with the ReLU that follows it fused in; pools and stray activations are the modules themselves.
    ``tapped(name)``: children whose output joins ``taps`` (encoder.forward's skip list)."""
    nn = torch.nn
    i = 0
    while i < len(children):
        name, child = children[i]
        last = name
        if isinstance(child, nn.Conv2d):
            if child.bias is not None or child.groups != 1 or child.kernel_size[0] != child.kernel_size[1]:
                raise BtsHipError("train: convolution %r is not built (bias-free, ungrouped, square only)" % (child,))
            x = conv2d(x, child.weight, padding=child.padding[0], dilation=child.dilation[0], stride=child.stride[0],
                       tag="enc")
        elif isinstance(child, nn.BatchNorm2d):
            fuse = (i + 1 < len(children) and isinstance(children[i + 1][1], nn.ReLU)
                    and not (tapped is not None and tapped(name)))
            x = _bn(x, child, relu=fuse)
            if fuse:
                i += 1
                last = children[i][0]
        elif isinstance(child, nn.ModuleDict):               # _DenseBlock
            fused = _dense_block(child, x) if FUSED_DENSE_BLOCKS else None
            if fused is not None:
                x = fused                                    # one autograd node, one NHWC buffer, no torch.cat
            else:                                            # generic graph: each layer sees the concat of all earlier ones
                feats = [x]
                for layer in child.values():
                    y = torch.cat(feats, 1) if len(feats) > 1 else feats[0]
                    feats.append(_run_children(list(layer.named_children()), y))   # norm1 relu1 conv1 norm2 relu2 conv2
                x = torch.cat(feats, 1)
        elif isinstance(child, nn.Sequential):               # _Transition
            x = _run_children(list(child.named_children()), x)
        else:
            x = child(x)
        if tapped is not None and tapped(last):
            taps.append(x)
        i += 1
    return x


def densenet_encoder_forward(enc, x):
    """encoder.forward (bts.py:327-338) for the DenseNet encoders in train() mode: same tap list, convolutions, norm
    layers and their gradients on libbts_hip.so."""
    ops._need(x, "train.encoder")
    begin_step()
    taps = [x]
    cur = x.float().contiguous(memory_format=torch.channels_last)
    with _deferred_batch_counts():
        _run_children(list(enc.base_model.named_children()), cur,
                      tapped=lambda name: any(fragment in name for fragment in enc.feat_names), taps=taps)
    return taps


# ------------------------------------------------------------------------------------------ encoder (ResNet / ResNeXt)
def _bottleneck(blk, x, tag):
    """torchvision Bottleneck.forward: 1x1 -> (grouped, strided) 3x3 -> 1x1, + identity, ReLU."""
    out = _bn(conv2d(x, blk.conv1.weight, tag=tag), blk.bn1, relu=True)
    c2 = blk.conv2
    out = _bn(conv2d(out, c2.weight, padding=c2.padding[0], dilation=c2.dilation[0], stride=c2.stride[0], groups=c2.groups,
                     tag=tag), blk.bn2, relu=True)
    out = _bn(conv2d(out, blk.conv3.weight, tag=tag), blk.bn3)
    if blk.downsample is not None:
        d = blk.downsample[0]
        x = _bn(conv2d(x, d.weight, stride=d.stride[0], tag=tag), blk.downsample[1])
    return F.relu(out + x)


def resnet_encoder_forward(enc, x):
    """encoder.forward (bts.py:327-338) for the ResNet-50/101 and ResNeXt-50/101 encoders in train() mode: taps
    [x, relu, layer1, layer2, layer3, layer4]; every convolution (incl. the grouped and strided ones) and norm layer and
    their gradients on libbts_hip.so, max-pool / residual add on PyTorch-ROCm kernels."""
    ops._need(x, "train.encoder")
    begin_step()
    m = enc.base_model
    cur = x.float().contiguous(memory_format=torch.channels_last)
    c1 = m.conv1
    with _deferred_batch_counts():
        cur = _bn(conv2d(cur, c1.weight, padding=c1.padding[0], stride=c1.stride[0], tag="enc"), m.bn1, relu=True)
        taps = [x, cur]
        cur = m.maxpool(cur)
        for li, layer in enumerate((m.layer1, m.layer2, m.layer3, m.layer4)):
            for blk in layer:
                cur = _bottleneck(blk, cur, "enc")
            taps.append(cur)
    return taps
