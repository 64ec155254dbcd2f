"""DenseNet and ResNet/ResNeXt encoders on the HIP conv kernel (NHWC, no MIOpen): ``DenseNetHip``, ``ResNetHip``.

The reference encoder is torchvision's DenseNet walked tap by tap (reference pytorch/bts.py:295-338);
it is the caller side of the decoder hot path.  This image has no MIOpen find-db for gfx950 (a first
pass JIT-compiles ~160 conv configs for minutes) and ATen's native conv path im2col's every layer,
so the same fp32-MFMA implicit-GEMM kernel that runs the decoder also runs the encoder:

  * a dense layer (norm1-relu1-conv1x1-norm2-relu2-conv3x3) is exactly the two-launch shape of the
    decoder's atrous_conv with dilation 1: BN+ReLU prologue on the gathered input, BN+ReLU epilogue,
    3x3 into a 48-channel slice;
  * torch.cat is free: every dense block is ONE preallocated NHWC buffer and each layer appends its
    growth channels in place (torchvision re-concatenates the whole prefix for every layer);
  * a transition (norm-relu-conv1x1-avgpool2) pools FIRST (the 1x1 conv commutes with the mean),
    quartering the conv;
  * the skip taps are written straight into the decoder's concat buffers (second conv/pool output),
    norm5+ReLU becomes the prologue of the decoder's first conv.

Weights come from the ordinary nn modules (same state_dict keys as torchvision) and are re-packed
lazily when they change.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from .workspace import WorkspaceCache, invalidate_packs, tensor_fingerprint, _generation


def _bn_vecs(bn: nn.BatchNorm2d, n_pad: int):
    s, b = ops.bn_affine(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)
    return ops.pad_vec(s, n_pad, 1.0), ops.pad_vec(b, n_pad, 0.0)


def _key(module: nn.Module):
    return (_generation[0],) + tensor_fingerprint(module)


class DenseNetHip:
    """Execution plan for ``encoders.densenet_features`` (torchvision ``DenseNet.features``)."""

    def __init__(self, features: nn.Sequential, key_module: Optional[nn.Module] = None):
        self.features = features
        self.key_module = key_module if key_module is not None else features   # whose tensors fingerprint the packs
        self._pack = None
        self._pack_key = None
        self._ws = WorkspaceCache(max_entries=8)
        f = features
        self.c_stem = f.conv0.out_channels
        self.blocks = [m for n, m in f.named_children() if n.startswith("denseblock")]
        self.transitions = [m for n, m in f.named_children() if n.startswith("transition")]
        self.growth = self.blocks[0].growth_rate
        self.c_mid = next(iter(self.blocks[0].values())).conv1.out_channels
        # channels entering / leaving each block
        self.c_in = [b.num_input_features for b in self.blocks]
        self.c_out = [b.num_input_features + len(b) * b.growth_rate for b in self.blocks]

    def __reduce__(self):          # deepcopy / pickle of the owning model: a fresh plan (no packs, no workspaces)
        return (DenseNetHip, (self.features, self.key_module))

    def bind(self, features: nn.Sequential, key_module: Optional[nn.Module] = None):
        """Point the plan at another instance of the same network -- a DataParallel replica, re-created on every
        forward (bts_test.py:91) -- without dropping packs or workspaces: the packs are fingerprinted on
        ``key_module`` (the SOURCE model's tensors), which the replica's freshly broadcast copies equal."""
        if features is not self.features:
            self.features = features
            self.blocks = [m for n, m in features.named_children() if n.startswith("denseblock")]
            self.transitions = [m for n, m in features.named_children() if n.startswith("transition")]
        self.key_module = key_module if key_module is not None else features
        return self

    # ------------------------------------------------------------------------- packing
    def packed(self):
        key = _key(self.key_module)
        if self._pack is not None and self._pack_key == key:
            return self._pack
        f = self.features
        P = {}
        w0, co0, _ = ops.pack_conv_weight(f.conv0.weight.detach(), c_in_ld=4)
        P["stem"] = dict(w=w0, e1=_bn_vecs(f.norm0, co0))
        P["blocks"] = []
        for blk in self.blocks:
            layers = []
            for layer in blk.values():
                cin = layer.conv1.in_channels
                w1, co1, cld1 = ops.pack_conv_weight(layer.conv1.weight.detach())
                w2, co2, cld2 = ops.pack_conv_weight(layer.conv2.weight.detach())
                layers.append(dict(cin=cin, w1=w1, pre=_bn_vecs(layer.norm1, cld1), e1=_bn_vecs(layer.norm2, co1), w2=w2))
            P["blocks"].append(layers)
        P["trans"] = []
        for tr in self.transitions:
            c = tr.conv.in_channels
            w, co, cld = ops.pack_conv_weight(tr.conv.weight.detach())
            s, b = _bn_vecs(tr.norm, c)
            P["trans"].append(dict(w=w, scale=s, shift=b, c_in=c, c_out=tr.conv.out_channels))
        P["norm5"] = _bn_vecs(f.norm5, ops.round_up(self.c_out[-1], 4))
        self._pack, self._pack_key = P, key
        return P

    # ----------------------------------------------------------------------- workspace
    def _workspace(self, B, H, W, device, slot=0):
        return self._ws.get((B, H, W, str(device), slot), lambda: self._alloc_workspace(B, H, W, device))

    def _alloc_workspace(self, B, H, W, device):
        n = [B * (H // s) * (W // s) for s in (1, 2, 4, 8, 16, 32)]

        def z(npix, c):
            return torch.zeros((npix, c), dtype=torch.float32, device=device)

        ws = dict(img=z(n[0], 4), mid=z(n[2], self.c_mid))
        # scratch for split-K of the M-starved deep layers (see bts_conv_desc.splitk_ws): 8 splits x [n/16^2, 512]
        ws["splitk"] = torch.empty(8 * n[4] * 512, dtype=torch.float32, device=device)
        for i in range(4):
            ws["blk%d" % i] = z(n[2 + i], self.c_out[i])
        for i in range(3):
            ws["pool%d" % i] = z(n[3 + i], self.c_out[i])
        return ws

    # ----------------------------------------------------------------------------- run
    def run(self, x: torch.Tensor, skip_dst: Optional[List[Optional[torch.Tensor]]] = None, slot: int = 0):
        """x [B,3,H,W] NCHW -> dict(blk3=[npix/32^2, C] NHWC view before norm5, norm5=(scale,shift), taps=...).

        ``skip_dst``: four optional [npix_s, C_s] NHWC views (decoder concat slots) receiving the taps
        relu0 (H/2), pool0 (H/4), transition1 (H/8), transition2 (H/16); missing ones are allocated.
        """
        ops._need(x, "DenseNetHip.run")
        B, C, H, W = x.shape
        if C != 3 or H % 32 or W % 32:
            raise ops.BtsHipError("DenseNetHip: expected [B,3,H,W] with H,W multiples of 32")
        dev = x.device
        P = self.packed()
        ws = self._workspace(B, H, W, dev, slot)
        skip_dst = list(skip_dst) if skip_dst is not None else [None] * 4
        hs = [H // s for s in (2, 4, 8, 16, 32)]
        wss = [W // s for s in (2, 4, 8, 16, 32)]
        taps_c = [self.c_stem, self.c_stem, self.c_in[1], self.c_in[2]]
        for i in range(4):
            if skip_dst[i] is None:
                skip_dst[i] = torch.empty((B * hs[i] * wss[i], taps_c[i]), dtype=torch.float32, device=dev)
        RELU = ops.ACT_RELU

        # stem: conv0 7x7/2 + norm0 + relu0 -> skip0 slot; pool0 -> block-1 prefix (+ skip1 slot)
        ops.nchw_to_nhwc(x, ws["img"][:, :3])
        ops.conv_forward(ws["img"], B, H, W, P["stem"]["w"], self.c_stem, 7, stride=2, pad=3, e1=P["stem"]["e1"], act=RELU,
                         y2d=skip_dst[0], tag="enc_stem", c_in_real=3)
        ops.maxpool3x3s2(skip_dst[0], B, hs[0], wss[0], ws["blk0"][:, :self.c_stem], skip_dst[1])

        for bi, layers in enumerate(P["blocks"]):
            h, w = hs[bi + 1], wss[bi + 1]
            buf = ws["blk%d" % bi]
            mid = ws["mid"][: B * h * w]
            for li, L in enumerate(layers):
                cin = L["cin"]
                ops.conv_forward(buf[:, :cin], B, h, w, L["w1"], self.c_mid, 1, pre=L["pre"], pre_relu=True, e1=L["e1"],
                                 act=RELU, y2d=mid, tag="enc_b%d_1x1" % (bi + 1), splitk_ws=ws["splitk"])
                ops.conv_forward(mid, B, h, w, L["w2"], self.growth, 3, y2d=buf[:, cin:cin + self.growth],
                                 tag="enc_b%d_3x3" % (bi + 1), splitk_ws=ws["splitk"])
            if bi < 3:
                T = P["trans"][bi]
                pooled = ws["pool%d" % bi]
                ops.bn_relu_avgpool2(buf, B, h, w, T["scale"], T["shift"], pooled)
                nxt = ws["blk%d" % (bi + 1)]
                ops.conv_forward(pooled, B, h // 2, w // 2, T["w"], T["c_out"], 1, y2d=nxt[:, :T["c_out"]],
                                 y2_2d=skip_dst[bi + 2] if bi < 2 else None, tag="enc_trans", splitk_ws=ws["splitk"])
        return dict(dense=ws["blk3"], norm5=P["norm5"], skips=skip_dst, B=B, H=H, W=W)

    def taps_nchw(self, x: torch.Tensor) -> List[torch.Tensor]:
        """The reference ``encoder.forward`` list [x, relu0, pool0, transition1, transition2, norm5] as NCHW
        tensors (bts.py:327-338), computed on the HIP path (used by tests / stand-alone encoder calls)."""
        r = self.run(x)
        B, H, W = r["B"], r["H"], r["W"]
        outs = [x]
        for i, s in enumerate(r["skips"]):
            outs.append(ops.nhwc_to_nchw(s, B, H // (2 << i), W // (2 << i)))
        d = r["dense"]
        sc, sh = r["norm5"]
        n5 = d * sc[: d.shape[1]] + sh[: d.shape[1]]          # norm5 is fused into the decoder; materialise it here only
        outs.append(ops.nhwc_to_nchw(n5.contiguous(), B, H // 32, W // 32))
        return outs


class ResNetHip:
    """Execution plan for ``encoders.ResNet`` (torchvision ResNet-50/101, ResNeXt-50 32x4d / -101 32x8d) in eval mode.

    Every convolution of a bottleneck is one launch of the conv kernel with its BatchNorm folded into the epilogue:
    conv1 1x1 (+BN+ReLU), conv2 3x3 (stride, groups; +BN+ReLU), the optional 1x1 strided downsample (+BN), and conv3
    1x1 whose epilogue adds the identity and applies the final ReLU (``bts_conv_desc.res``).  ResNeXt's 32-group 3x3
    runs as ONE launch over channel bundles (``bts_conv_desc.n_bundles``): consecutive groups packed block-diagonally
    into >= 32-channel bundles, so the MFMA tiles stay full at the price of (32 / channels-per-group)x redundant
    FLOPs on the two shallow stages only.  The taps the decoder needs (relu, layer1..3; reference bts.py:318-338) are
    written straight into its concat buffers; layer4's output is the decoder's dense input."""

    def __init__(self, model: nn.Module, key_module: Optional[nn.Module] = None):
        self.model = model
        self.key_module = key_module if key_module is not None else model
        self._pack = None
        self._pack_key = None
        self._ws = WorkspaceCache(max_entries=8)
        self.layers = [model.layer1, model.layer2, model.layer3, model.layer4]
        self.c_stem = model.conv1.out_channels
        self.c_out = [l[-1].conv3.out_channels for l in self.layers]
        self.width = [l[0].conv2.out_channels for l in self.layers]

    def __reduce__(self):
        return (ResNetHip, (self.model, self.key_module))

    def bind(self, model: nn.Module, key_module: Optional[nn.Module] = None):
        """See DenseNetHip.bind."""
        if model is not self.model:
            self.model = model
            self.layers = [model.layer1, model.layer2, model.layer3, model.layer4]
        self.key_module = key_module if key_module is not None else model
        return self

    def packed(self):
        key = _key(self.key_module)
        if self._pack is not None and self._pack_key == key:
            return self._pack
        m = self.model
        w0, co0, _ = ops.pack_conv_weight(m.conv1.weight.detach(), c_in_ld=4)
        P = dict(stem=dict(w=w0, e1=_bn_vecs(m.bn1, co0)), layers=[])
        for layer in self.layers:
            blocks = []
            for blk in layer:
                w1, co1, _ = ops.pack_conv_weight(blk.conv1.weight.detach())
                g = blk.conv2.groups
                if g > 1:
                    w2, nb, cb = ops.pack_grouped_conv_weight(blk.conv2.weight.detach(), g)
                else:
                    w2, _, _ = ops.pack_conv_weight(blk.conv2.weight.detach())
                    nb, cb = 1, blk.conv2.out_channels
                w3, co3, _ = ops.pack_conv_weight(blk.conv3.weight.detach())
                width = blk.conv2.out_channels
                b = dict(w1=w1, e1=_bn_vecs(blk.bn1, co1), w2=w2, nb=nb, cb=cb, cg=blk.conv2.in_channels // g,
                         e2=_bn_vecs(blk.bn2, ops.round_up(width, 32)), w3=w3, e3=_bn_vecs(blk.bn3, co3),
                         stride=blk.conv2.stride[0], width=width, c_out=blk.conv3.out_channels, down=None)
                if blk.downsample is not None:
                    wd, cod, _ = ops.pack_conv_weight(blk.downsample[0].weight.detach())
                    b["down"] = dict(w=wd, e1=_bn_vecs(blk.downsample[1], cod), stride=blk.downsample[0].stride[0])
                blocks.append(b)
            P["layers"].append(blocks)
        self._pack, self._pack_key = P, key
        return P

    def _workspace(self, B, H, W, device, slot=0):
        return self._ws.get((B, H, W, str(device), slot), lambda: self._alloc_workspace(B, H, W, device))

    def _alloc_workspace(self, B, H, W, device):
        n = [B * (H // s) * (W // s) for s in (1, 2, 4, 8, 16, 32)]

        def z(npix, c):
            return torch.zeros((npix, c), dtype=torch.float32, device=device)

        ws = dict(img=z(n[0], 4), x0=z(n[2], self.c_stem))
        ws["splitk"] = torch.empty(8 * n[3] * 512, dtype=torch.float32, device=device)
        for li in range(4):
            px_in, px = n[2 + max(li - 1, 0)] if li else n[2], n[2 + li]
            ws["t1_%d" % li] = z(max(px_in, px), self.width[li])
            ws["t2_%d" % li] = z(px, self.width[li])
            ws["idn_%d" % li] = z(px, self.c_out[li])
            ws["a_%d" % li] = z(px, self.c_out[li])
            ws["b_%d" % li] = z(px, self.c_out[li])
        return ws

    def run(self, x: torch.Tensor, skip_dst: Optional[List[Optional[torch.Tensor]]] = None, slot: int = 0):
        """x [B,3,H,W] NCHW -> dict(dense=[npix/32^2, c_out[3]] NHWC (layer4, already ReLU'd), norm5=None, skips=...).
        ``skip_dst``: four optional NHWC views receiving relu (H/2), layer1 (H/4), layer2 (H/8), layer3 (H/16)."""
        ops._need(x, "ResNetHip.run")
        B, C, H, W = x.shape
        if C != 3 or H % 32 or W % 32:
            raise ops.BtsHipError("ResNetHip: expected [B,3,H,W] with H,W multiples of 32")
        dev = x.device
        P = self.packed()
        ws = self._workspace(B, H, W, dev, slot)
        skip_dst = list(skip_dst) if skip_dst is not None else [None] * 4
        hs = [H // s for s in (2, 4, 8, 16, 32)]
        wss = [W // s for s in (2, 4, 8, 16, 32)]
        taps_c = [self.c_stem] + self.c_out[:3]
        for i in range(4):
            if skip_dst[i] is None:
                skip_dst[i] = torch.empty((B * hs[i] * wss[i], taps_c[i]), dtype=torch.float32, device=dev)
        RELU = ops.ACT_RELU
        sk = ws["splitk"]

        ops.nchw_to_nhwc(x, ws["img"][:, :3])
        ops.conv_forward(ws["img"], B, H, W, P["stem"]["w"], self.c_stem, 7, stride=2, pad=3, e1=P["stem"]["e1"], act=RELU,
                         y2d=skip_dst[0], tag="enc_stem", c_in_real=3)
        ops.maxpool3x3s2(skip_dst[0], B, hs[0], wss[0], ws["x0"])
        cur, h, w = ws["x0"], hs[1], wss[1]
        for li, blocks in enumerate(P["layers"]):
            tag = "enc_l%d" % (li + 1)
            for bi, b in enumerate(blocks):
                s = b["stride"]
                ho, wo = h // s, w // s
                t1 = ws["t1_%d" % li][: B * h * w]
                t2 = ws["t2_%d" % li][: B * ho * wo]
                ops.conv_forward(cur, B, h, w, b["w1"], b["width"], 1, e1=b["e1"], act=RELU, y2d=t1, tag=tag + "_1x1",
                                 splitk_ws=sk)
                if b["nb"] > 1:
                    ops.conv_forward(t1, B, h, w, b["w2"], b["cb"], 3, stride=s, pad=1, c_in_ld=b["cb"], e1=b["e2"], act=RELU,
                                     y2d=t2, n_bundles=b["nb"], tag=tag + "_g3x3", c_in_real=b["cg"])
                else:
                    ops.conv_forward(t1, B, h, w, b["w2"], b["width"], 3, stride=s, pad=1, e1=b["e2"], act=RELU, y2d=t2,
                                     tag=tag + "_3x3", splitk_ws=sk)
                if b["down"] is not None:
                    idn = ws["idn_%d" % li][: B * ho * wo]
                    ops.conv_forward(cur, B, h, w, b["down"]["w"], b["c_out"], 1, stride=b["down"]["stride"], pad=0,
                                     e1=b["down"]["e1"], y2d=idn, tag=tag + "_down", splitk_ws=sk)
                else:
                    idn = cur
                out = ws["a_%d" % li] if (bi % 2 == 0) else ws["b_%d" % li]
                last = bi == len(blocks) - 1
                ops.conv_forward(t2, B, ho, wo, b["w3"], b["c_out"], 1, e1=b["e3"], act=RELU, y2d=out, res2d=idn,
                                 y2_2d=skip_dst[li + 1] if (last and li < 3) else None, tag=tag + "_1x1", splitk_ws=sk)
                cur, h, w = out, ho, wo
        return dict(dense=cur, norm5=None, skips=skip_dst, B=B, H=H, W=W)

    def taps_nchw(self, x: torch.Tensor) -> List[torch.Tensor]:
        """The reference ``encoder.forward`` list [x, relu, layer1, layer2, layer3, layer4] as NCHW tensors
        (bts.py:327-338), computed on the HIP path."""
        r = self.run(x)
        B, H, W = r["B"], r["H"], r["W"]
        outs = [x]
        for i, s in enumerate(r["skips"]):
            outs.append(ops.nhwc_to_nchw(s, B, H // (2 << i), W // (2 << i)))
        outs.append(ops.nhwc_to_nchw(r["dense"], B, H // 32, W // 32))
        return outs
