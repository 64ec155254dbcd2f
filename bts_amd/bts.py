"""Drop-in model plugin: the names, constructor signatures, attribute tree and state_dict keys of
the reference ``pytorch/bts.py`` with the decoder hot path running on hand-written HIP kernels.

The reference's callers load the model file as a plugin --
``for k, v in vars(__import__(args.model_name)).items(): vars()[k] = v`` (bts_test.py:70-76,
bts_eval.py:72-78, bts_main.py:63-76) -- then ``BtsModel(params=args)``, ``load_state_dict``,
``.eval()``, ``.cuda()``, ``model(image, focal)`` -> 6 NCHW fp32 tensors (bts_test.py:90-138).
This module exports the same symbols (``BtsModel, encoder, bts, atrous_conv, upconv,
reduction_1x1, local_planar_guidance, silog_loss, depth_l1_loss, weights_init_xavier,
bn_init_as_tf``); parameters live in ordinary nn.Conv2d / nn.BatchNorm2d modules so keys and
shapes are identical to the reference (110 decoder entries), and are re-packed lazily into the
kernels' layouts after ``load_state_dict``.

Inside ``bts.forward`` everything is NHWC: every convolution of the decoder is one launch of the
fp32-MFMA implicit-GEMM kernel with its BN/ReLU/ELU fused, concatenations are channel slices of
preallocated buffers (no torch.cat), each reduction_1x1 stack is one kernel, each LPG one kernel.
In ``train()`` mode (bts_main.py) the same modules build an autograd graph instead -- batch-statistic BN and every
convolution's forward / input gradient / weight gradient on the HIP kernels, see ``bts_amd/train.py``.
There is no PyTorch/CPU fallback: CPU tensors or a missing libbts_hip.so raise.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn as nn

from . import ops, train
from ._lib import BtsHipError
from . import encoders
from .encoders import build_base_model
from .workspace import PackCache, WorkspaceCache, tensor_fingerprint
from . import workspace as _workspace
from .plan import PlanCache
from . import _lib as _lib_mod


# ----------------------------------------------------------------- helpers kept from the reference API
def bn_init_as_tf(m):
    """bts.py:26-31."""
    if isinstance(m, nn.BatchNorm2d):
        m.track_running_stats = True
        m.eval()
        m.affine = True
        m.requires_grad = True


def weights_init_xavier(m):
    """bts.py:34-38."""
    if isinstance(m, nn.Conv2d):
        torch.nn.init.xavier_uniform_(m.weight)
        if m.bias is not None:
            torch.nn.init.zeros_(m.bias)


class silog_loss(nn.Module):
    """Scale-invariant log loss, same value as reference bts.py:41-48 (plain torch; training callers only)."""

    def __init__(self, variance_focus):
        super().__init__()
        self.variance_focus = variance_focus

    def forward(self, depth_est, depth_gt, mask):
        log_ratio = depth_est[mask].log() - depth_gt[mask].log()
        second_moment, first_moment = log_ratio.pow(2).mean(), log_ratio.mean()
        return 10.0 * torch.sqrt(second_moment - self.variance_focus * first_moment * first_moment)


class depth_l1_loss(nn.Module):
    """Asymmetric L1 depth loss, same value as reference bts.py:50-63: over-estimates are weighted by
    ``inbalance_to_closer``; the mean runs over ALL masked pixels."""

    def __init__(self, inbalance_to_closer):
        super().__init__()
        self.inbalance_to_closer = inbalance_to_closer

    def forward(self, depth_est, depth_gt, mask):
        err = depth_est[mask] - depth_gt[mask]
        if self.inbalance_to_closer == 1:
            return err.abs().mean()
        weighted = torch.where(err > 0, self.inbalance_to_closer * err, -err)
        return weighted.sum() / err.numel()


def _bn_vecs(bn: nn.BatchNorm2d, n_pad: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Eval-mode BN as padded (scale, shift) vectors."""
    s, b = ops.bn_affine(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)
    return ops.pad_vec(s, n_pad, 1.0), ops.pad_vec(b, n_pad, 0.0)


_param_key = tensor_fingerprint      # re-pack when load_state_dict / .cuda() / an optimizer touches a module's tensors


def _nhwc_in(x: torch.Tensor, c_pad_to: int = 4) -> Tuple[torch.Tensor, int, int, int, int]:
    """NCHW module-boundary tensor -> [npix, C_ld] NHWC buffer (zero pad channels)."""
    ops._need(x, "forward")
    B, C, H, W = x.shape
    buf = torch.zeros((B * H * W, ops.round_up(C, c_pad_to)), dtype=torch.float32, device=x.device)
    ops.nchw_to_nhwc(x, buf[:, :C])
    return buf, B, C, H, W


# ----------------------------------------------------------------------------------------- modules
class atrous_conv(nn.Sequential):
    """bts.py:65-80.  Same sub-module tree (atrous_conv.first_bn / aconv_sequence.{1,2,4})."""

    def __init__(self, in_channels, out_channels, dilation, apply_bn_first=True):
        super().__init__()
        mid = 2 * out_channels
        branch = nn.Sequential()                     # registration order fixes the checkpoint keys
        if apply_bn_first:
            branch.add_module('first_bn', nn.BatchNorm2d(in_channels, eps=1.1e-5, momentum=0.01))
        branch.add_module('aconv_sequence', nn.Sequential(
            nn.ReLU(),                                                                     # .0
            nn.Conv2d(in_channels, mid, kernel_size=1, bias=False),                        # .1
            nn.BatchNorm2d(mid, momentum=0.01),                                            # .2
            nn.ReLU(),                                                                     # .3
            nn.Conv2d(mid, out_channels, kernel_size=3, padding=dilation, dilation=dilation, bias=False)))   # .4
        self.atrous_conv = branch
        self.dilation = dilation
        self.apply_bn_first = apply_bn_first
        self._packs = PackCache(self)        # shared with DataParallel replicas (bts_amd/workspace.py)

    def packed(self):
        seq = self.atrous_conv.aconv_sequence

        def build():
            w1, co1, k1 = ops.pack_conv_weight(seq[1].weight.detach())
            w2, co2, k2 = ops.pack_conv_weight(seq[4].weight.detach())
            pre = _bn_vecs(self.atrous_conv.first_bn, k1) if self.apply_bn_first else None   # k1 == c_in_ld
            e1 = _bn_vecs(seq[2], co1)
            return dict(w1=w1, w2=w2, pre=pre, e1=e1, c_mid=seq[1].out_channels, c_out=seq[4].out_channels,
                        c_in=seq[1].in_channels)
        return self._packs.get(seq[1].weight.device, build)

    def run_nhwc(self, x2d, B, h, w, mid2d, y2d, splitk_ws=None):
        """Two launches: [BN]+ReLU -> 1x1 -> BN -> ReLU (mid), then dilated 3x3 into the y2d slice.  ``splitk_ws``:
        scratch that lets the launches split K in single-frame mode (fill_frames 1 / 2); at the default setting
        the H/8 maps fill the chip and never split."""
        p = self.packed()
        ops.conv_forward(x2d, B, h, w, p["w1"], p["c_mid"], 1, c_in_ld=p["c_in"], pre=p["pre"], pre_relu=True,
                         e1=p["e1"], act=ops.ACT_RELU, y2d=mid2d, tag="aspp", splitk_ws=splitk_ws)
        ops.conv_forward(mid2d, B, h, w, p["w2"], p["c_out"], 3, dil=self.dilation, y2d=y2d, tag="aspp", splitk_ws=splitk_ws)

    def forward(self, x):
        if self.training:
            return train.atrous_forward(self, x)          # batch-statistic BN + autograd graph (bts_amd/train.py)
        xin, B, C, h, w = _nhwc_in(x)
        p = self.packed()
        if C % 4:
            raise BtsHipError("atrous_conv: in_channels must be a multiple of 4")
        mid = torch.empty((B * h * w, p["c_mid"]), dtype=torch.float32, device=x.device)
        out = torch.empty((B * h * w, p["c_out"]), dtype=torch.float32, device=x.device)
        self.run_nhwc(xin, B, h, w, mid, out)
        return ops.nhwc_to_nchw(out, B, h, w)


class upconv(nn.Module):
    """bts.py:83-94: nearest x2 (folded into the conv's gather) -> conv3x3 -> ELU."""

    def __init__(self, in_channels, out_channels, ratio=2):
        super().__init__()
        self.elu = nn.ELU()          # parameter-free; kept so the module tree prints like the reference's
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1, bias=False)
        self.ratio = ratio
        self._packs = PackCache(self)
        self._packs_taps = PackCache(self)

    def packed(self):
        """ratio 2: ("subpix", four 2x2 kernels, ops.pack_upconv_subpixel) or -- ``self.tap_gemm`` -- ("taps", the nine taps
        side by side as one 1x1 weight, ops.pack_upconv_taps); plain 3x3 packing otherwise."""
        def build():
            if self.ratio == 2 and self.tap_gemm:
                return ("taps",) + ops.pack_upconv_taps(self.conv.weight.detach())
            if self.ratio == 2:
                return ("subpix",) + ops.pack_upconv_subpixel(self.conv.weight.detach())
            return ("plain",) + ops.pack_conv_weight(self.conv.weight.detach())
        cache = self._packs_taps if (self.ratio == 2 and self.tap_gemm) else self._packs
        return cache.get(self.conv.weight.device, build)

    # nearest-2x + 3x3 as ONE 1x1 convolution with the nine taps side by side plus a tap-sum pass (9 tap-products per
    # source pixel) instead of four 2x2 parity convolutions (16): set by the decoder for upconv5, whose 11x38 x 2208-channel
    # source map is where the GEMM dominates and the extra pass is small
    tap_gemm = False

    def run_nhwc(self, x2d, B, h, w, y2d, taps_buf=None, e2=None, pre=None, pre_relu=False, c_in_real=None, splitk_ws=None,
                 tag="decoder_upconv"):
        """ELU(conv3x3(nearest2x(x))) [-> e2 affine] from the NHWC view x2d [B*h*w, C] into y2d [B*2h*2w, cout]."""
        form, wp, n_or_pad, c_in_ld = self.packed()
        cout = self.conv.out_channels
        cin = c_in_real if c_in_real is not None else self.conv.in_channels
        if form == "taps":
            if taps_buf is None:
                taps_buf = torch.empty((B * h * w, 9 * cout), dtype=torch.float32, device=x2d.device)
            ops.conv_forward(x2d, B, h, w, wp, 9 * cout, 1, c_in_ld=c_in_ld, pre=pre, pre_relu=pre_relu, y2d=taps_buf[:, :9 * cout],
                             tag=tag, c_in_real=cin, splitk_ws=splitk_ws, algo_flops=2.0 * 36 * B * h * w * cout * cin)
            return ops.upconv_combine(taps_buf, B, h, w, cout, y2d, act=ops.ACT_ELU, e2=e2, tag=tag + "_sum")
        return ops.conv_forward(x2d, B, h, w, wp, cout, 3, dil=1, up=2, act=ops.ACT_ELU, e2=e2, pre=pre, pre_relu=pre_relu,
                                y2d=y2d, tag=tag, c_in_real=cin, subpixel=True, splitk_ws=splitk_ws)

    def forward(self, x):
        if self.ratio not in (1, 2):
            raise BtsHipError("upconv: ratio %r not built (1 or 2)" % (self.ratio,))
        if self.training:
            return train.upconv_forward(self, x)
        xin, B, C, h, w = _nhwc_in(x)
        cout = self.conv.out_channels
        r = self.ratio
        y2d = torch.empty((B * h * r * w * r, cout), dtype=torch.float32, device=x.device)
        if r == 2:
            self.run_nhwc(xin, B, h, w, y2d, tag="conv")
        else:
            ops.conv_forward(xin, B, h, w, self.packed()[1], cout, 3, dil=1, up=1, act=ops.ACT_ELU, y2d=y2d)
        return ops.nhwc_to_nchw(y2d, B, h * r, w * r)


class reduction_1x1(nn.Sequential):
    """bts.py:97-136.  Same ``reduc`` sub-module names (inter_<in>_<out>, plane_params / final)."""

    def __init__(self, num_in_filters, num_out_filters, max_depth, is_final=False):
        super().__init__()
        self.max_depth = max_depth
        self.is_final = is_final
        self.sigmoid = nn.Sigmoid()
        self.c_in, self.c_first_out = num_in_filters, num_out_filters
        stack = nn.Sequential()
        for cin, cout in ops.reduc_chain(num_in_filters, num_out_filters):     # widths halve until < 8 (bts.py:105-122)
            if cout > 0:
                stack.add_module('inter_{}_{}'.format(cin, cout),
                                 nn.Sequential(nn.Conv2d(cin, cout, kernel_size=1, bias=False), nn.ELU()))
            elif is_final:
                stack.add_module('final', nn.Sequential(nn.Conv2d(cin, 1, kernel_size=1, bias=False), nn.Sigmoid()))
            else:
                stack.add_module('plane_params', nn.Conv2d(cin, 3, kernel_size=1, bias=False))
        self.reduc = stack
        self._packs = PackCache(self)

    def packed(self) -> torch.Tensor:
        ws = [m.weight.detach() for m in self.reduc.modules() if isinstance(m, nn.Conv2d)]
        return self._packs.get(ws[0].device, lambda: ops.pack_reduc_weights(ws))

    def run_nhwc(self, x2d, out, normalize):
        ops.reduc_forward_nhwc(x2d, self.c_in, self.c_first_out, self.packed(), self.max_depth, self.is_final,
                               normalize, out)

    def run_lpg_nhwc(self, x2d, B, h, w, upratio, depth_scaled, ds_out, abs_min):
        """This reduction + normalize + LPG + /max_depth (+ downsampled plane) as ONE launch (bts.py:249-256)."""
        ops.reduc_lpg_forward(x2d, B, h, w, self.c_in, self.c_first_out, self.packed(), self.max_depth, upratio,
                              depth_scaled, ds_out=ds_out, abs_min=abs_min)

    def forward(self, net):
        if self.training:
            return train.reduction_forward(self, net)
        xin, B, C, h, w = _nhwc_in(net)
        if self.is_final:
            out = torch.empty((B, 1, h, w), dtype=torch.float32, device=net.device)
            self.run_nhwc(xin[:, :C], out, False)
            return out
        out = torch.empty((B * h * w, 4), dtype=torch.float32, device=net.device)
        self.run_nhwc(xin[:, :C], out, False)
        return ops.nhwc_to_nchw(out, B, h, w)


class local_planar_guidance(nn.Module):
    """bts.py:138-173.  ``abs_min`` is kept as a device scalar (bts_main.py:484-486 reads it)."""

    def __init__(self, upratio):
        super().__init__()
        self.upratio = float(upratio)
        self.abs_min = None

    def forward(self, plane_eq, focal):
        tops = ops.torch_ops()
        if tops is not None:
            # the torch operator: validation, device guard and current stream in the C++ shell, autograd registered on it
            # (bts_hip::lpg / bts_hip::lpg_backward -- the pair the reference registers as LocalPlanarGuidance /
            # LocalPlanarGuidanceGrad, local_planar_guidance.cc:31-72, 234-239)
            ops._need(plane_eq, "local_planar_guidance")
            box = []
            ops._op(lambda: box.append(tops.lpg(plane_eq, int(self.upratio))))
            depth, am = box[0]
            self.abs_min = am.detach()
            return depth
        am = torch.empty((), dtype=torch.float32, device=plane_eq.device)
        if torch.is_grad_enabled() and plane_eq.requires_grad:
            depth = ops.LpgFunction.apply(plane_eq, int(self.upratio), am)      # native backward (bts_lpg_bwd_f32)
        else:
            depth = ops.lpg_forward(plane_eq, int(self.upratio), abs_min=am)
        self.abs_min = am
        return depth


# ----------------------------------------------------------------------------------------- decoder
class bts(nn.Module):
    """bts.py:175-293.  Same attribute names / state_dict; forward runs NHWC on HIP end to end."""

    def __init__(self, params, feat_out_channels, num_features=512):
        super().__init__()
        self.params = params
        self.feat_out_channels = f = list(feat_out_channels)
        self.num_features = nf = num_features
        md = params.max_depth

        def bn(c):
            return nn.BatchNorm2d(c, eps=1.1e-5, momentum=0.01)

        def conv_elu(cin, cout):
            return nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1, bias=False), nn.ELU())

        cat4 = nf // 2 + f[2]                       # channels of concat4 = [upconv4 | skip2]
        q = nf // 4                                 # ASPP branch width
        # (attribute name, module): registration order == state_dict order of the reference decoder (bts.py:180-221)
        plan = [
            ("upconv5", upconv(f[4], nf)), ("bn5", bn(nf)), ("conv5", conv_elu(nf + f[3], nf)),
            ("upconv4", upconv(nf, nf // 2)), ("bn4", bn(nf // 2)), ("conv4", conv_elu(cat4, nf // 2)), ("bn4_2", bn(nf // 2)),
            ("daspp_3", atrous_conv(nf // 2, q, 3, apply_bn_first=False)),
            ("daspp_6", atrous_conv(cat4 + 1 * q, q, 6)), ("daspp_12", atrous_conv(cat4 + 2 * q, q, 12)),
            ("daspp_18", atrous_conv(cat4 + 3 * q, q, 18)), ("daspp_24", atrous_conv(cat4 + 4 * q, q, 24)),
            ("daspp_conv", conv_elu(nf // 2 + 5 * q, q)),
            ("reduc8x8", reduction_1x1(q, q, md)), ("lpg8x8", local_planar_guidance(8)),
            ("upconv3", upconv(q, q)), ("bn3", bn(q)), ("conv3", conv_elu(q + f[1] + 1, q)),
            ("reduc4x4", reduction_1x1(q, nf // 8, md)), ("lpg4x4", local_planar_guidance(4)),
            ("upconv2", upconv(q, nf // 8)), ("bn2", bn(nf // 8)), ("conv2", conv_elu(nf // 8 + f[0] + 1, nf // 8)),
            ("reduc2x2", reduction_1x1(nf // 8, nf // 16, md)), ("lpg2x2", local_planar_guidance(2)),
            ("upconv1", upconv(nf // 8, nf // 16)),
            ("reduc1x1", reduction_1x1(nf // 16, nf // 32, md, is_final=True)),
            ("conv1", conv_elu(nf // 16 + 4, nf // 16)),
            ("get_depth", nn.Sequential(nn.Conv2d(nf // 16, 1, 3, 1, 1, bias=False), nn.Sigmoid())),
        ]
        for name, module in plan:
            setattr(self, name, module)
        # small, wide maps (11x38 x 2208 -> 512, 22x76 x 512 -> 256): 9 tap-products per source pixel instead of the
        # sub-pixel form's 16; measured 43.3 -> 42.3 -> 41.9 ms/step.  upconv3 (44x152, 256 -> 128) is neutral: its tap
        # buffer costs what its GEMM saves; upconv2/1 stay sub-pixel on the halo-tile kernel
        self.upconv5.tap_gemm = True
        self.upconv4.tap_gemm = True
        self._packs = PackCache(self)
        self._bufs = WorkspaceCache(max_entries=8)      # per-shape NHWC workspaces; shared with replicas, graph-pinnable
        self.fill_frames = None                         # launch declaration when the decoder is called on its own
        self.conv_precision = "fp32"                    # (BtsModel.forward opens its own scope: see BtsModel.fill_frames)

    # ------------------------------------------------------------------ weight packing (lazy)
    _OWN = ("bn5", "conv5", "bn4", "conv4", "bn4_2", "daspp_conv", "bn3", "conv3", "bn2", "conv2", "conv1", "get_depth")

    def packed(self):
        return self._packs.get(self.conv5[0].weight.device, self._build_pack,
                               key_modules=lambda origin: [getattr(origin, n) for n in self._OWN])

    def _build_pack(self):
        nf = self.num_features
        f = self.feat_out_channels
        P = {}
        P["conv5"] = ops.pack_conv_weight(self.conv5[0].weight.detach())
        P["conv4"] = ops.pack_conv_weight(self.conv4[0].weight.detach())
        # daspp_conv reads buffer order [d3,d6,d12,d18,d24,iconv4]; reference order is [iconv4,d3,...] (bts.py:246)
        n_d = 5 * (nf // 4)
        perm = torch.cat([torch.arange(nf // 2, nf // 2 + n_d), torch.arange(0, nf // 2)])
        P["daspp_conv"] = ops.pack_conv_weight(self.daspp_conv[0].weight.detach(), perm=perm)
        # conv3 / conv2 / conv1 read their last 1 / 1 / 4 input channels (depth_8x8_scaled_ds, depth_4x4_scaled_ds;
        # reduc1x1 + the three depth maps: bts.py:260, 274, 287) from dense planes -- bts_conv_desc.tail_planes -- so
        # the packed K axis is [buffer channels | 4 tail slots]
        for name, m in (("conv3", self.conv3), ("conv2", self.conv2), ("conv1", self.conv1)):
            w = m[0].weight.detach()
            n_tail = 4 if name == "conv1" else 1
            c_main = w.shape[1] - n_tail
            if c_main % 4:
                raise BtsHipError("bts: %s has %d feature channels in front of its depth planes; the planar-tail "
                                  "convolution needs a multiple of 4" % (name, c_main))
            P[name] = ops.pack_conv_weight(w, c_in_ld=c_main + 4)
        P["bn5"] = _bn_vecs(self.bn5, ops.round_up(nf, 32))
        P["bn4"] = _bn_vecs(self.bn4, ops.round_up(nf // 2, 32))
        P["bn4_2"] = _bn_vecs(self.bn4_2, ops.round_up(nf // 2, 32))
        P["bn3"] = _bn_vecs(self.bn3, ops.round_up(nf // 4, 32))
        P["bn2"] = _bn_vecs(self.bn2, ops.round_up(nf // 8, 32))
        P["get_depth"] = self.get_depth[0].weight.detach().float().contiguous()
        return P

    # ------------------------------------------------------------------ NHWC workspace (per shape)
    def _workspace(self, B: int, H: int, W: int, device, slot: int = 0) -> Dict[str, torch.Tensor]:
        key = (B, H, W, str(device), slot)
        return self._bufs.get(key, lambda: self._alloc_workspace(B, H, W, device))

    def _alloc_workspace(self, B: int, H: int, W: int, device) -> Dict[str, torch.Tensor]:
        nf, f = self.num_features, self.feat_out_channels
        n32, n16, n8, n4, n2, n1 = [B * (H // s) * (W // s) for s in (32, 16, 8, 4, 2, 1)]

        def z(n, c):
            return torch.zeros((n, c), dtype=torch.float32, device=device)

        r4 = ops.round_up
        ws = dict(
            f5=z(n32, r4(f[4], 4)),
            taps5=torch.empty((n32, 9 * nf), dtype=torch.float32, device=device),   # upconv5 as a tap GEMM: nine taps x nf
            taps4=torch.empty((n16, 9 * (nf // 2)), dtype=torch.float32, device=device),
            cat5=z(n16, r4(nf + f[3], 4)),                       # [upconv5 | skip3]
            iconv5=z(n16, nf),
            x8=z(n8, nf // 2 + f[2] + 5 * (nf // 4) + nf // 2),  # [upconv4 | skip2 | d3 d6 d12 d18 d24 | iconv4]
            mid=z(n8, nf // 2),
            daspp_feat=z(n8, nf // 4),
            cat3=z(n4, nf // 4 + f[1]),                          # [upconv3 | skip1]; depth_8x8_scaled_ds is the plane ds8
            ds8=torch.zeros(n4, dtype=torch.float32, device=device),
            iconv3=z(n4, nf // 4),
            cat2=z(n2, nf // 8 + f[0]),                          # [upconv2 | skip0]; depth_4x4_scaled_ds is the plane ds4
            ds4=torch.zeros(n2, dtype=torch.float32, device=device),
            iconv2=z(n2, nf // 8),
            cat1=z(n1, nf // 16),                                # upconv1; reduc1x1 / d2 / d4 / d8 are read as planes
            # scratch for split-K of under-filled launches (bts_conv_desc.splitk_ws): 8 splits x [n16, nf]
            splitk=torch.empty(8 * n16 * nf, dtype=torch.float32, device=device),
        )
        return ws

    def skip_slots(self, ws):
        """The four NHWC channel slices of the decoder's concat buffers that hold the encoder taps
        (skip0 @H/2, skip1 @H/4, skip2 @H/8, skip3 @H/16): an NHWC producer can write them in place."""
        nf, f = self.num_features, self.feat_out_channels
        return [ws["cat2"][:, nf // 8:nf // 8 + f[0]], ws["cat3"][:, nf // 4:nf // 4 + f[1]],
                ws["x8"][:, nf // 2:nf // 2 + f[2]], ws["cat5"][:, nf:nf + f[3]]]

    def forward(self, features, focal):
        """bts.forward(features, focal), bts.py:223-293: NCHW encoder taps in, the reference's 6-tuple out."""
        if not ops.launch_config_active() and len(features) == 6 and isinstance(features[5], torch.Tensor):
            with ops.model_launch_config(self, features[5].shape[0]):       # called on its own: the decoder's declaration
                return self._forward(features, focal)
        return self._forward(features, focal)

    def _forward(self, features, focal):
        if len(features) != 6:
            raise BtsHipError("bts.forward: expected the encoder's 6-element tap list, got %d" % len(features))
        skip0, skip1, skip2, skip3, dense = features[1], features[2], features[3], features[4], features[5]
        ops._need(dense, "bts.forward")
        B = dense.shape[0]
        H, W = dense.shape[2] * 32, dense.shape[3] * 32
        f = self.feat_out_channels
        # host-side shape contract: kernels index these buffers with the sizes derived here
        for i, (t, s) in enumerate(zip((skip0, skip1, skip2, skip3, dense), (2, 4, 8, 16, 32))):
            ops._need(t, "bts.forward")
            want = (B, f[i], H // s, W // s)
            if tuple(t.shape) != want:
                raise BtsHipError("bts.forward: features[%d] has shape %s, expected %s (H, W multiples of 32; "
                                  "channels %s as in bts.py:300-323)" % (i + 1, tuple(t.shape), want, f))
        if focal is not None and self.params.dataset == 'kitti' and focal.numel() != B:
            raise BtsHipError("bts.forward: focal must have one entry per frame (%d), got %d" % (B, focal.numel()))
        if self.training:
            return train.decoder_forward(self, features, focal.to(dense.device) if isinstance(focal, torch.Tensor) else focal)
        ws = self._workspace(B, H, W, dense.device)
        # boundary: NCHW encoder taps -> NHWC channel slices (dense_features = ReLU(features[5]), bts.py:225)
        ops.nchw_to_nhwc(dense, ws["f5"][:, :f[4]], relu=True)
        for src, dst in zip((skip0, skip1, skip2, skip3), self.skip_slots(ws)):
            ops.nchw_to_nhwc(src, dst)
        return self.forward_nhwc(ws, B, H, W, focal, ws["f5"], None, False)

    def forward_nhwc(self, ws, B, H, W, focal, dense2d, dense_pre, dense_relu, outs=None, abs_mins=None):
        """The decoder proper on NHWC buffers.  ``ws``: this module's workspace with the four skip slots
        already filled; ``dense2d``: the 1/32-resolution features [npix, C>=f[4]] and the prologue
        (affine, relu) still to be applied to them (norm5 + ReLU when the encoder is fused in).
        ``outs``: optional 6 preallocated contiguous result tensors (e.g. batch slices of full-batch tensors).
        ``abs_mins``: optional float32 [3] tensor that receives the three LPG ``abs_min`` scalars (8x8, 4x4, 2x2)."""
        if self.training:
            raise BtsHipError("bts.forward_nhwc is the fused inference path; train() mode goes through bts.forward "
                              "(bts_amd/train.py)")
        dev = dense2d.device
        nf, f = self.num_features, self.feat_out_channels
        md = float(self.params.max_depth)
        P = self.packed()
        h16, w16, h8, w8, h4, w4, h2, w2 = H // 16, W // 16, H // 8, W // 8, H // 4, W // 4, H // 2, W // 2
        ELU = ops.ACT_ELU

        def conv(name_w, x2d, hh, ww, cout, y2d=None, y_nchw=None, up=1, e2=None, c_in_real=None, pre=None,
                 pre_relu=False, tail=None):
            wp = P[name_w] if isinstance(name_w, str) else name_w
            return ops.conv_forward(x2d, B, hh, ww, wp[0], cout, 3, dil=1, up=up, act=ELU, e2=e2, pre=pre, pre_relu=pre_relu,
                                    y2d=y2d, y_nchw=y_nchw, tag="decoder_upconv" if up == 2 else "decoder_conv",
                                    c_in_real=c_in_real, subpixel=(up == 2), splitk_ws=ws["splitk"], tail_planes=tail)

        # H/16 and H/8 trunk (bts.py:226-235)
        sk = ws["splitk"]
        self.upconv5.run_nhwc(dense2d, B, H // 32, W // 32, ws["cat5"][:, :nf], taps_buf=ws.get("taps5"), e2=P["bn5"],
                              pre=dense_pre, pre_relu=dense_relu, c_in_real=f[4], splitk_ws=sk)
        conv("conv5", ws["cat5"], h16, w16, nf, y2d=ws["iconv5"], c_in_real=nf + f[3])
        x8 = ws["x8"]
        c_cat4 = nf // 2 + f[2]
        o_d = c_cat4                              # first ASPP output slot
        o_i4 = c_cat4 + 5 * (nf // 4)             # iconv4 slot
        self.upconv4.run_nhwc(ws["iconv5"], B, h16, w16, x8[:, :nf // 2], taps_buf=ws.get("taps4"), e2=P["bn4"], splitk_ws=sk)
        conv("conv4", x8[:, :c_cat4], h8, w8, nf // 2, y2d=x8[:, o_i4:o_i4 + nf // 2], e2=P["bn4_2"])

        # dense ASPP (bts.py:237-247): each branch reads a prefix of x8 and appends its 128 channels
        q = nf // 4
        self.daspp_3.run_nhwc(x8[:, o_i4:o_i4 + nf // 2], B, h8, w8, ws["mid"], x8[:, o_d:o_d + q], ws["splitk"])
        self.daspp_6.run_nhwc(x8[:, :o_d + q], B, h8, w8, ws["mid"], x8[:, o_d + q:o_d + 2 * q], ws["splitk"])
        self.daspp_12.run_nhwc(x8[:, :o_d + 2 * q], B, h8, w8, ws["mid"], x8[:, o_d + 2 * q:o_d + 3 * q], ws["splitk"])
        self.daspp_18.run_nhwc(x8[:, :o_d + 3 * q], B, h8, w8, ws["mid"], x8[:, o_d + 3 * q:o_d + 4 * q], ws["splitk"])
        self.daspp_24.run_nhwc(x8[:, :o_d + 4 * q], B, h8, w8, ws["mid"], x8[:, o_d + 4 * q:o_d + 5 * q], ws["splitk"])
        conv("daspp_conv", x8[:, o_d:], h8, w8, q, y2d=ws["daspp_feat"])

        am_buf = abs_mins if abs_mins is not None else torch.empty(3, dtype=torch.float32, device=dev)
        am_it = iter((am_buf[0], am_buf[1], am_buf[2]))

        def am():
            return next(am_it)

        def out_tensor(i, c):
            if outs is not None:
                t = outs[i]
                if tuple(t.shape) != (B, c, H, W) or not t.is_contiguous():
                    raise BtsHipError("forward_nhwc: outs[%d] must be contiguous [%d,%d,%d,%d]" % (i, B, c, H, W))
                return t
            return torch.empty((B, c, H, W), dtype=torch.float32, device=dev)

        # 8x8 scale (bts.py:249-256)
        depth_8x8_scaled = out_tensor(0, 1)
        a8 = am()
        c3 = ws["cat3"]
        self.reduc8x8.run_lpg_nhwc(ws["daspp_feat"], B, h8, w8, 8, depth_8x8_scaled, ws["ds8"], a8)   # ONE launch
        self.lpg8x8.abs_min = a8

        # H/4 (bts.py:258-270)
        self.upconv3.run_nhwc(ws["daspp_feat"], B, h8, w8, c3[:, :q], e2=P["bn3"], splitk_ws=sk)
        conv("conv3", c3, h4, w4, q, y2d=ws["iconv3"], c_in_real=q + f[1] + 1, tail=[ws["ds8"]])
        depth_4x4_scaled = out_tensor(1, 1)
        a4 = am()
        c2 = ws["cat2"]
        self.reduc4x4.run_lpg_nhwc(ws["iconv3"], B, h4, w4, 4, depth_4x4_scaled, ws["ds4"], a4)
        self.lpg4x4.abs_min = a4

        # H/2 (bts.py:272-283)
        self.upconv2.run_nhwc(ws["iconv3"], B, h4, w4, c2[:, :nf // 8], e2=P["bn2"], splitk_ws=sk)
        conv("conv2", c2, h2, w2, nf // 8, y2d=ws["iconv2"], c_in_real=nf // 8 + f[0] + 1, tail=[ws["ds4"]])
        depth_2x2_scaled = out_tensor(2, 1)
        a2 = am()
        self.reduc2x2.run_lpg_nhwc(ws["iconv2"], B, h2, w2, 2, depth_2x2_scaled, None, a2)
        self.lpg2x2.abs_min = a2

        # full resolution (bts.py:285-291)
        c1 = ws["cat1"]
        n16c = nf // 16
        self.upconv1.run_nhwc(ws["iconv2"], B, h2, w2, c1, splitk_ws=sk)
        reduc1x1 = out_tensor(3, 1)
        self.reduc1x1.run_nhwc(c1, reduc1x1, False)
        iconv1 = out_tensor(5, n16c)
        # concat1 = cat[upconv1, reduc1x1, depth_2x2, depth_4x4, depth_8x8] (bts.py:287): the four maps are read in place
        conv("conv1", c1, H, W, n16c, y_nchw=iconv1, c_in_real=n16c + 4,
             tail=[reduc1x1, depth_2x2_scaled, depth_4x4_scaled, depth_8x8_scaled])
        fo = None
        if self.params.dataset == 'kitti':
            fo = focal.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
        final_depth = ops.get_depth_forward(iconv1, P["get_depth"], md, fo, out=out_tensor(4, 1))

        return depth_8x8_scaled, depth_4x4_scaled, depth_2x2_scaled, reduc1x1, final_depth, iconv1


class encoder(nn.Module):
    """bts.py:295-338.  Same ``base_model`` tree / tap names; weights come from the checkpoint
    (``pretrained=True`` needs the network and torchvision, neither available here)."""

    TAPS = {  # encoder name -> (module-name fragments that are tapped, channels of the five taps), bts.py:300-323
        'densenet121_bts': (('relu0', 'pool0', 'transition1', 'transition2', 'norm5'), [64, 64, 128, 256, 1024]),
        'densenet161_bts': (('relu0', 'pool0', 'transition1', 'transition2', 'norm5'), [96, 96, 192, 384, 2208]),
    }
    RESNET_TAPS = (('relu', 'layer1', 'layer2', 'layer3', 'layer4'), [64, 256, 512, 1024, 2048])

    def __init__(self, params):
        super().__init__()
        self.params = params
        self.base_model = build_base_model(params.encoder)
        names, channels = self.TAPS.get(params.encoder, self.RESNET_TAPS)
        self.feat_names, self.feat_out_channels = list(names), list(channels)

    def forward(self, x):
        """[x, tap@1/2, tap@1/4, tap@1/8, tap@1/16, tap@1/32]: run the backbone's children in order and keep the outputs
        of those whose name contains a tap fragment (bts.py:327-338; 'fc'/'avgpool' children are skipped)."""
        taps, cur = [x], x
        for child_name, child in self.base_model.named_children():
            if 'fc' in child_name or 'avgpool' in child_name:
                continue
            cur = child(cur)
            if any(fragment in child_name for fragment in self.feat_names):
                taps.append(cur)
        return taps


_SIDE_STREAMS: Dict[str, list] = {}


class BtsModel(nn.Module):
    """bts.py:341-349.  Same constructor and forward signature.

    In eval mode with a GPU input the whole forward is native: the encoder (DenseNet-121/161, ResNet-50/101,
    ResNeXt-50/101) runs on the HIP conv kernel (bts_amd.encoder_hip) writing its taps straight into the decoder's
    NHWC concat buffers.  ``native_encoder = False`` runs the torch encoder modules instead (NCHW taps handed to the
    HIP decoder).  train() mode: see bts_amd/train.py."""

    def __init__(self, params):
        super(BtsModel, self).__init__()
        self.encoder = encoder(params)
        self.decoder = bts(params, self.encoder.feat_out_channels, params.bts_size)
        self.native_encoder = True          # set False to force the torch encoder (A/B, debugging)
        self.sub_batches = 4                # concurrent sub-batches (own HIP stream + workspace each); 1 = off
                                            # (MI355X, B=16: 1 -> 54.7, 2 -> 48.4, 4 -> 47.6, 8 -> 51.5 ms/step)
        self.fill_frames = None             # frames per launch this model declares to the library (bts_conv_desc.fill_frames:
                                            # sizes split-K and the tile family, so it changes fp32 summation order).
                                            # None = by the batch of each call, in three classes (ops.auto_fill_frames:
                                            # B <= 2 -> 2, the latency setting of bts_test.py's B=1 loop; B <= 11 -> 8;
                                            # else 16); an int pins it, and a frame's bits then never depend on the batch
        self.conv_precision = "fp32"        # "fp32": fp32-input MFMA; "bf16x3": fp32 emulated on the bf16 matrix cores
        self.output_buffers = None          # optional: up to six preallocated contiguous result tensors ([B,1,H,W] x 5,
                                            # [B,32,H,W]); entries given (not None) receive the results of the native eval
                                            # forward IN PLACE instead of fresh tensors -- e.g. views of a persistent
                                            # all-gather send buffer (dist.DepthGather.outputs): no pack copy per step
        self.use_plans = False              # True: record each (shape, slot) forward once, replay it with ONE library
                                            # call per forward afterwards (bts_amd/plan.py, bts_plan_run): the eager
                                            # B=1 loop of bts_test.py:127-147 without ~120 ctypes crossings per frame
        self._plans = PlanCache()
        self._fp_state = [0, None, -1]      # workspace.tensor_fingerprint: [structure token, cached tensor list, its token]
        self._origin = [self]               # reaches DataParallel replicas through replicate()'s shallow __dict__ copy
        self._enc_plans = {}                # device -> DenseNetHip / ResNetHip (packs + workspaces), shared with replicas

    def _native_ok(self, x):
        return (self.native_encoder and not self.training and isinstance(x, torch.Tensor) and x.is_cuda
                and isinstance(self.encoder.base_model, (nn.Sequential, encoders.ResNet)))

    def _forward_native(self, x, focal, slot, outs=None, abs_mins=None):
        if self.use_plans and ops._trace is None and not _lib_mod.is_recording():
            r = self._plans.forward(self, x, focal, slot, outs)
            if abs_mins is not None:
                dec = self.decoder
                torch.stack([dec.lpg8x8.abs_min, dec.lpg4x4.abs_min, dec.lpg2x2.abs_min], out=abs_mins)
            return r
        return self._forward_native_eager(x, focal, slot, outs, abs_mins)

    def _forward_native_eager(self, x, focal, slot, outs=None, abs_mins=None):
        from .encoder_hip import DenseNetHip, ResNetHip
        base = self.encoder.base_model
        src_base = self._origin[0].encoder.base_model      # a replica's packs are fingerprinted on the source model
        cls = ResNetHip if isinstance(base, encoders.ResNet) else DenseNetHip
        plan = self._enc_plans.get(str(x.device))
        if plan is None or type(plan) is not cls or getattr(plan, "_src", None) is not src_base:
            plan = cls(base, key_module=src_base)
            plan._src = src_base
            self._enc_plans[str(x.device)] = plan
        plan.bind(base, key_module=src_base)
        plan._ws.max_entries = max(plan._ws.max_entries, 2 * int(self.sub_batches))
        B, _, H, W = x.shape
        dec = self.decoder
        ws = dec._workspace(B, H, W, x.device, slot)
        r = plan.run(x.float(), dec.skip_slots(ws), slot=slot)
        # DenseNet: norm5 + ReLU become the prologue of the decoder's first conv; ResNet: layer4 is already ReLU'd
        return dec.forward_nhwc(ws, B, H, W, focal, r["dense"], r["norm5"], r["norm5"] is not None, outs=outs, abs_mins=abs_mins)

    def _result_tensors(self, B, H, W, dev):
        """The six result tensors of one native eval forward: ``output_buffers`` entries where given, fresh otherwise."""
        nf16 = self.decoder.num_features // 16
        given = list(self.output_buffers or [])
        res = []
        for i, c in enumerate((1, 1, 1, 1, 1, nf16)):
            buf = given[i] if i < len(given) else None
            if buf is None:
                buf = torch.empty((B, c, H, W), dtype=torch.float32, device=dev)
            elif (tuple(buf.shape) != (B, c, H, W) or buf.dtype != torch.float32 or buf.device != dev or not buf.is_contiguous()):
                raise BtsHipError("BtsModel.output_buffers[%d] must be a contiguous float32 [%d,%d,%d,%d] tensor on %s, got %s %s on %s"
                                  % (i, B, c, H, W, dev, tuple(buf.shape), buf.dtype, buf.device))
            res.append(buf)
        return res

    def _apply(self, fn, recurse=True):
        # .cuda() / .to() / .float(): buffers are REPLACED by new tensor objects -- the cached fingerprint list is stale
        self._fp_state[0] += 1
        return super()._apply(fn, recurse)

    def load_state_dict(self, state_dict, strict=True, assign=False):
        if assign:
            self._fp_state[0] += 1          # assign=True swaps the parameter objects themselves
        return super().load_state_dict(state_dict, strict=strict, assign=assign)

    def train(self, mode: bool = True):
        # a mode switch is where a training loop hands weights over to evaluation (bts_main.py:193-275 evaluates every
        # eval_freq steps): age every packed copy, whatever way the optimiser wrote the parameters (workspace.py)
        if mode != self.training:
            _workspace.invalidate_packs()
        return super().train(mode)

    def forward(self, x, focal):
        # this model's launch declaration (fill_frames, precision) holds for every kernel launch of the call: a
        # thread-local scope, so two models -- or two DataParallel replicas -- never see each other's setting
        if isinstance(x, torch.Tensor) and x.dim() == 4 and not ops.launch_config_active():
            with ops.model_launch_config(self, x.shape[0]):
                return self._forward(x, focal)
        return self._forward(x, focal)

    def _forward(self, x, focal):
        if self.training and self.native_encoder and isinstance(x, torch.Tensor) and x.is_cuda:
            # training step: encoder + decoder as one autograd graph on the HIP kernels (bts_amd/train.py)
            enc_fwd = train.resnet_encoder_forward if isinstance(self.encoder.base_model, encoders.ResNet) \
                else train.densenet_encoder_forward
            with train.model_step():
                return self.decoder(enc_fwd(self.encoder, x), focal)
        if not self._native_ok(x):
            skip_feat = self.encoder(x)
            return self.decoder(skip_feat, focal)
        B, _, H, W = x.shape
        S = int(self.sub_batches)
        while S > 1 and B % S:              # largest divisor of B not above the requested count
            S -= 1
        if S <= 1:
            return self._forward_native(x, focal, 0, outs=self._result_tensors(B, H, W, x.device) if self.output_buffers else None)
        # Frames are independent in eval mode (bts.py:223-293 has no cross-sample op), so the batch runs as S
        # concurrent sub-batches, each on its own stream with its own NHWC workspace and writing its slice of
        # the full-batch outputs: the under-filled launches of one (deep encoder layers, tile-quantisation
        # tails) overlap the other's kernels.  Per-frame results are bit-identical to the single-stream path.
        dev = x.device
        full = self._result_tensors(B, H, W, dev)
        streams = _SIDE_STREAMS.setdefault(str(dev), [])        # per device, process-wide (not module state: a module
        while len(streams) < S:                                  # holding Stream objects cannot be deep-copied / saved)
            streams.append(torch.cuda.Stream(dev))
        dec_ws = self.decoder._bufs                              # one forward needs S workspaces alive at once
        dec_ws.max_entries = max(dec_ws.max_entries, 2 * S)
        cur = torch.cuda.current_stream(dev)
        b = B // S
        mins = torch.empty((S, 3), dtype=torch.float32, device=dev)      # per-stream abs_min triples, written by the kernels
        focal_d = focal.to(device=dev) if isinstance(focal, torch.Tensor) else focal
        for i in range(S):
            st = streams[i]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                lo, hi = i * b, (i + 1) * b
                self._forward_native(x[lo:hi], focal_d[lo:hi] if isinstance(focal_d, torch.Tensor) else focal_d, i,
                                     outs=[t[lo:hi] for t in full], abs_mins=mins[i])
        for i in range(S):
            cur.wait_stream(streams[i])
        am = mins.min(dim=0).values                                   # abs_min over the whole batch (bts.py:167); NaN propagates
        self.decoder.lpg8x8.abs_min, self.decoder.lpg4x4.abs_min, self.decoder.lpg2x2.abs_min = am[0], am[1], am[2]
        return tuple(full)
