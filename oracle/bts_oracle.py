"""CPU ORACLE for the BTS decoder hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A plain-torch (CPU, fp32) functional restatement of the reference decoder
``/root/reference/pytorch/bts.py`` for eval mode.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product path (``bts_amd``) never does and fails loudly when
its HIP library is missing.

Parity status: PINNED by import-generated goldens.  The reference ships no
tests or golden vectors for this path, so the oracle is pinned
against outputs of the reference itself, generated in the build container by
``tests/golden/gen_golden.py`` (which imports ``/root/reference/pytorch/bts.py``
with a ``.cuda()``->identity shim) and committed under ``tests/golden/``.
``tests/test_oracle_golden.py`` asserts this file reproduces them bit-exactly
(same torch CPU kernels, same op order).  Third-party arithmetic (conv2d,
batch_norm, elu, sigmoid, sin/cos) lives in PyTorch itself (reference "tested
under PyTorch 1.2.0", pytorch/README.md:8; here torch 2.10 CPU): unpinned
except through those goldens.

Every function cites the reference lines it restates.  The op ORDER of the
reference is kept (separate mul/add, where-clamps, true divisions) so results
are bit-identical to the reference on the same torch build.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- LPG
def lpg_forward(plane_eq: Tensor, upratio: int) -> Tuple[Tensor, Tensor]:
    """local_planar_guidance.forward, bts.py:149-173.

    plane_eq [B,4,h,w] -> depth [B,h*k,w*k]; also returns abs_min (bts.py:167).
    u varies along columns, v along rows (bts.py:142-143,157-161).
    """
    k = int(upratio)
    B, _, h, w = plane_eq.shape
    pe = torch.repeat_interleave(plane_eq, k, 2)
    pe = torch.repeat_interleave(pe, k, 3)
    n1, n2, n3, n4 = pe[:, 0], pe[:, 1], pe[:, 2], pe[:, 3]
    # the reference builds u, v on the CPU and moves them with .cuda() every call (bts.py:157,160); `.to` is
    # the device-agnostic spelling (a no-op on CPU tensors)
    u = torch.arange(k).reshape(1, 1, k).float().repeat(B, h * k, w).to(plane_eq.device)
    u = (u - (float(k) - 1) * 0.5) / float(k)
    v = torch.arange(k).reshape(1, k, 1).float().repeat(B, h, w * k).to(plane_eq.device)
    v = (v - (float(k) - 1) * 0.5) / float(k)
    divided = n1 * u + n2 * v + n3
    abs_min = torch.abs(divided).min()
    eps = 1e-3
    dummy_eps = torch.ones_like(divided) * eps
    divided = torch.where((divided > 0) & (divided < eps), dummy_eps, divided)
    divided = torch.where((divided < 0) & (divided > -eps), -dummy_eps, divided)
    return n4 / divided, abs_min


def lpg_denominator(plane_eq: Tensor, upratio: int) -> Tensor:
    """The un-clamped denominator n1*u+n2*v+n3 (bts.py:166); used by tests to mask
    near-singular pixels (|den| < 2e-3) where relative error is meaningless."""
    k = int(upratio)
    B, _, h, w = plane_eq.shape
    pe = torch.repeat_interleave(torch.repeat_interleave(plane_eq, k, 2), k, 3)
    u = torch.arange(k).reshape(1, 1, k).float().repeat(B, h * k, w)
    u = (u - (float(k) - 1) * 0.5) / float(k)
    v = torch.arange(k).reshape(1, k, 1).float().repeat(B, h, w * k)
    v = (v - (float(k) - 1) * 0.5) / float(k)
    return pe[:, 0] * u + pe[:, 1] * v + pe[:, 2]


# --------------------------------------------------------------------- reduction
def reduction_forward(x: Tensor, weights: Sequence[Tensor], max_depth: float,
                      is_final: bool) -> Tensor:
    """reduction_1x1.forward, bts.py:124-136.

    ``weights``: the chain's 1x1 conv weights in order ([cout,cin,1,1], no bias).
    All but the last layer are conv+ELU (bts.py:116-119); the last is
    ``plane_params`` (no activation, bts.py:112-113) or ``final`` conv+Sigmoid
    (bts.py:108-110).
    """
    net = x
    for w in weights[:-1]:
        net = F.elu(F.conv2d(net, w))
    net = F.conv2d(net, weights[-1])
    if is_final:
        return torch.sigmoid(net)
    theta = torch.sigmoid(net[:, 0, :, :]) * math.pi / 3
    phi = torch.sigmoid(net[:, 1, :, :]) * math.pi * 2
    dist = torch.sigmoid(net[:, 2, :, :]) * max_depth
    n1 = torch.mul(torch.sin(theta), torch.cos(phi)).unsqueeze(1)
    n2 = torch.mul(torch.sin(theta), torch.sin(phi)).unsqueeze(1)
    n3 = torch.cos(theta).unsqueeze(1)
    n4 = dist.unsqueeze(1)
    return torch.cat([n1, n2, n3, n4], dim=1)


# ------------------------------------------------------------------------- ASPP
def _bn_eval(x: Tensor, p: Dict[str, Tensor], prefix: str, eps: float, training: bool = False) -> Tensor:
    """nn.BatchNorm2d(momentum=0.01): running statistics in eval mode; with ``training`` the batch statistics
    (and the running buffers in ``p`` are updated in place, as module.train() does)."""
    return F.batch_norm(x, p[prefix + ".running_mean"], p[prefix + ".running_var"],
                        p[prefix + ".weight"], p[prefix + ".bias"], training, 0.01, eps)


def atrous_forward(x: Tensor, p: Dict[str, Tensor], prefix: str, dilation: int,
                   apply_bn_first: bool, training: bool = False) -> Tensor:
    """atrous_conv.forward, bts.py:65-80 (eval-mode BN unless ``training``).

    [first_bn eps 1.1e-5] -> ReLU -> conv1x1 -> BN(eps 1e-5) -> ReLU -> conv3x3 dilated.
    ``p`` maps '<prefix>.atrous_conv.*' keys to tensors.
    """
    a = prefix + ".atrous_conv"
    if apply_bn_first:
        x = _bn_eval(x, p, a + ".first_bn", 1.1e-5, training)
    x = F.relu(x)
    x = F.conv2d(x, p[a + ".aconv_sequence.1.weight"])
    x = _bn_eval(x, p, a + ".aconv_sequence.2", 1e-5, training)
    x = F.relu(x)
    return F.conv2d(x, p[a + ".aconv_sequence.4.weight"], padding=dilation, dilation=dilation)


# ----------------------------------------------------------------- other blocks
def upconv_forward(x: Tensor, w: Tensor, ratio: int = 2) -> Tensor:
    """upconv.forward, bts.py:90-94: nearest x2 -> conv3x3 -> ELU."""
    up = F.interpolate(x, scale_factor=ratio, mode="nearest")
    return F.elu(F.conv2d(up, w, padding=1))


def _reduc_weights(p: Dict[str, Tensor], name: str) -> List[Tensor]:
    pre = name + ".reduc."
    keys = [k for k in p.keys() if k.startswith(pre) and k.endswith(".weight")]
    # module order == insertion order of the state dict (inter_* ... then plane_params/final)
    return [p[k] for k in keys]


def decoder_forward(p: Dict[str, Tensor], features: Sequence[Optional[Tensor]], focal: Tensor,
                    max_depth: float, dataset: str,
                    want_intermediates: bool = False, training: bool = False):
    """bts.forward, bts.py:223-293 (eval mode; ``training`` = module.train(): batch-statistic BN, autograd
    through every op when the tensors in ``p`` / ``features`` require grad).  ``p``: decoder state dict (no prefix).

    Returns the reference's 6-tuple; with ``want_intermediates`` also a dict of
    named internals used by per-kernel parity tests.
    """
    skip0, skip1, skip2, skip3 = features[1], features[2], features[3], features[4]
    dense_features = F.relu(features[5])
    upconv5 = upconv_forward(dense_features, p["upconv5.conv.weight"])
    upconv5 = _bn_eval(upconv5, p, "bn5", 1.1e-5, training)
    concat5 = torch.cat([upconv5, skip3], dim=1)
    iconv5 = F.elu(F.conv2d(concat5, p["conv5.0.weight"], padding=1))

    upconv4 = upconv_forward(iconv5, p["upconv4.conv.weight"])
    upconv4 = _bn_eval(upconv4, p, "bn4", 1.1e-5, training)
    concat4 = torch.cat([upconv4, skip2], dim=1)
    iconv4 = F.elu(F.conv2d(concat4, p["conv4.0.weight"], padding=1))
    iconv4 = _bn_eval(iconv4, p, "bn4_2", 1.1e-5, training)

    daspp_3 = atrous_forward(iconv4, p, "daspp_3", 3, False, training)
    concat4_2 = torch.cat([concat4, daspp_3], dim=1)
    daspp_6 = atrous_forward(concat4_2, p, "daspp_6", 6, True, training)
    concat4_3 = torch.cat([concat4_2, daspp_6], dim=1)
    daspp_12 = atrous_forward(concat4_3, p, "daspp_12", 12, True, training)
    concat4_4 = torch.cat([concat4_3, daspp_12], dim=1)
    daspp_18 = atrous_forward(concat4_4, p, "daspp_18", 18, True, training)
    concat4_5 = torch.cat([concat4_4, daspp_18], dim=1)
    daspp_24 = atrous_forward(concat4_5, p, "daspp_24", 24, True, training)
    concat4_daspp = torch.cat([iconv4, daspp_3, daspp_6, daspp_12, daspp_18, daspp_24], dim=1)
    daspp_feat = F.elu(F.conv2d(concat4_daspp, p["daspp_conv.0.weight"], padding=1))

    def lpg_scale(reduc, k):
        normal = F.normalize(reduc[:, :3, :, :], 2, 1)
        dist = reduc[:, 3, :, :]
        plane_eq = torch.cat([normal, dist.unsqueeze(1)], 1)
        depth, abs_min = lpg_forward(plane_eq, k)
        return plane_eq, depth.unsqueeze(1) / max_depth, abs_min

    reduc8x8 = reduction_forward(daspp_feat, _reduc_weights(p, "reduc8x8"), max_depth, False)
    plane_eq_8x8, depth_8x8_scaled, am8 = lpg_scale(reduc8x8, 8)
    depth_8x8_scaled_ds = F.interpolate(depth_8x8_scaled, scale_factor=0.25, mode="nearest")

    upconv3 = upconv_forward(daspp_feat, p["upconv3.conv.weight"])
    upconv3 = _bn_eval(upconv3, p, "bn3", 1.1e-5, training)
    concat3 = torch.cat([upconv3, skip1, depth_8x8_scaled_ds], dim=1)
    iconv3 = F.elu(F.conv2d(concat3, p["conv3.0.weight"], padding=1))

    reduc4x4 = reduction_forward(iconv3, _reduc_weights(p, "reduc4x4"), max_depth, False)
    plane_eq_4x4, depth_4x4_scaled, am4 = lpg_scale(reduc4x4, 4)
    depth_4x4_scaled_ds = F.interpolate(depth_4x4_scaled, scale_factor=0.5, mode="nearest")

    upconv2 = upconv_forward(iconv3, p["upconv2.conv.weight"])
    upconv2 = _bn_eval(upconv2, p, "bn2", 1.1e-5, training)
    concat2 = torch.cat([upconv2, skip0, depth_4x4_scaled_ds], dim=1)
    iconv2 = F.elu(F.conv2d(concat2, p["conv2.0.weight"], padding=1))

    reduc2x2 = reduction_forward(iconv2, _reduc_weights(p, "reduc2x2"), max_depth, False)
    plane_eq_2x2, depth_2x2_scaled, am2 = lpg_scale(reduc2x2, 2)

    upconv1 = upconv_forward(iconv2, p["upconv1.conv.weight"])
    reduc1x1 = reduction_forward(upconv1, _reduc_weights(p, "reduc1x1"), max_depth, True)
    concat1 = torch.cat([upconv1, reduc1x1, depth_2x2_scaled, depth_4x4_scaled, depth_8x8_scaled], dim=1)
    iconv1 = F.elu(F.conv2d(concat1, p["conv1.0.weight"], padding=1))
    final_depth = max_depth * torch.sigmoid(F.conv2d(iconv1, p["get_depth.0.weight"], padding=1))
    if dataset == "kitti":
        final_depth = final_depth * focal.view(-1, 1, 1, 1).float() / 715.0873

    outs = (depth_8x8_scaled, depth_4x4_scaled, depth_2x2_scaled, reduc1x1, final_depth, iconv1)
    if not want_intermediates:
        return outs
    inter = dict(iconv5=iconv5, concat4=concat4, iconv4=iconv4, daspp_3=daspp_3, daspp_6=daspp_6,
                 daspp_12=daspp_12, daspp_18=daspp_18, daspp_24=daspp_24, daspp_feat=daspp_feat,
                 reduc8x8=reduc8x8, plane_eq_8x8=plane_eq_8x8, iconv3=iconv3, reduc4x4=reduc4x4,
                 plane_eq_4x4=plane_eq_4x4, iconv2=iconv2, reduc2x2=reduc2x2,
                 plane_eq_2x2=plane_eq_2x2, upconv1=upconv1,
                 abs_min_8x8=am8, abs_min_4x4=am4, abs_min_2x2=am2)
    return outs, inter


def silog_loss(depth_est: Tensor, depth_gt: Tensor, mask: Tensor, variance_focus: float) -> Tensor:
    """silog_loss.forward, bts.py:41-48: d = log(est[mask]) - log(gt[mask]); sqrt(mean(d^2) - vf*mean(d)^2) * 10."""
    d = torch.log(depth_est[mask]) - torch.log(depth_gt[mask])
    return torch.sqrt((d ** 2).mean() - variance_focus * (d.mean() ** 2)) * 10.0


def state_from_numpy(state_np) -> Dict[str, Tensor]:
    return {k: torch.from_numpy(v.copy()) if hasattr(v, "shape") and v.shape != () else torch.tensor(int(v))
            for k, v in state_np.items()}
