"""Synthetic, regenerable parameters and inputs for the BTS decoder hot path.

Everything here is drawn from one NumPy ``PCG64(seed)`` stream in a fixed
(sorted-key) order, so the GPU box regenerates bit-identical weights and
inputs from a seed instead of shipping ~80 MB of tensors.

Shapes follow the reference decoder's state_dict (``pytorch/bts.py:175-221``):
conv weights are xavier-uniform (as ``weights_init_xavier``, bts.py:34-38),
BatchNorm gets non-trivial affine + running statistics so eval-mode BN is a
real per-channel affine, not an identity.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Sequence, Tuple

import numpy as np

# (encoder name) -> feat_out_channels, bts.py:300-323
ENCODER_CHANNELS = {
    "densenet121_bts": [64, 64, 128, 256, 1024],
    "densenet161_bts": [96, 96, 192, 384, 2208],
    "resnet50_bts": [64, 256, 512, 1024, 2048],
    "resnet101_bts": [64, 256, 512, 1024, 2048],
    "resnext50_bts": [64, 256, 512, 1024, 2048],
    "resnext101_bts": [64, 256, 512, 1024, 2048],
}

KITTI_FOCALS = (721.5377, 718.856, 718.3351, 707.0912, 707.0493)
NYU_FOCAL = 518.8579


def _reduc_chain(num_in: int, num_out: int, is_final: bool) -> List[Tuple[str, int, int]]:
    """Layer list of reduction_1x1 (bts.py:105-122): (module key, cin, cout)."""
    layers = []
    while num_out >= 4:
        if num_out < 8:
            if is_final:
                layers.append(("final.0", num_in, 1))
            else:
                layers.append(("plane_params", num_in, 3))
            break
        layers.append(("inter_{}_{}.0".format(num_in, num_out), num_in, num_out))
        num_in = num_out
        num_out = num_out // 2
    return layers


def reduc_chain_channels(num_in: int, num_out: int, is_final: bool) -> List[int]:
    """Channel chain, e.g. (128,128,False) -> [128,128,64,32,16,8,3]."""
    ls = _reduc_chain(num_in, num_out, is_final)
    return [ls[0][1]] + [l[2] for l in ls]


def decoder_param_shapes(feat: Sequence[int], nf: int = 512) -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict keys/shapes of reference ``bts`` (bts.py:175-221), in module order.

    BatchNorm entries expand to weight/bias/running_mean/running_var/num_batches_tracked.
    """
    d: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def conv(name, cout, cin, k):
        d[name + ".weight"] = (cout, cin, k, k)

    def bn(name, c):
        d[name + ".weight"] = (c,)
        d[name + ".bias"] = (c,)
        d[name + ".running_mean"] = (c,)
        d[name + ".running_var"] = (c,)
        d[name + ".num_batches_tracked"] = ()

    def atrous(name, cin, cout, first_bn):
        if first_bn:
            bn(name + ".atrous_conv.first_bn", cin)
        conv(name + ".atrous_conv.aconv_sequence.1", cout * 2, cin, 1)
        bn(name + ".atrous_conv.aconv_sequence.2", cout * 2)
        conv(name + ".atrous_conv.aconv_sequence.4", cout, cout * 2, 3)

    def reduc(name, num_in, num_out, is_final=False):
        for key, cin, cout in _reduc_chain(num_in, num_out, is_final):
            conv(name + ".reduc." + key, cout, cin, 1)

    conv("upconv5.conv", nf, feat[4], 3)
    bn("bn5", nf)
    conv("conv5.0", nf, nf + feat[3], 3)
    conv("upconv4.conv", nf // 2, nf, 3)
    bn("bn4", nf // 2)
    conv("conv4.0", nf // 2, nf // 2 + feat[2], 3)
    bn("bn4_2", nf // 2)
    atrous("daspp_3", nf // 2, nf // 4, False)
    atrous("daspp_6", nf // 2 + nf // 4 + feat[2], nf // 4, True)
    atrous("daspp_12", nf + feat[2], nf // 4, True)
    atrous("daspp_18", nf + nf // 4 + feat[2], nf // 4, True)
    atrous("daspp_24", nf + nf // 2 + feat[2], nf // 4, True)
    conv("daspp_conv.0", nf // 4, nf + nf // 2 + nf // 4, 3)
    reduc("reduc8x8", nf // 4, nf // 4)
    conv("upconv3.conv", nf // 4, nf // 4, 3)
    bn("bn3", nf // 4)
    conv("conv3.0", nf // 4, nf // 4 + feat[1] + 1, 3)
    reduc("reduc4x4", nf // 4, nf // 8)
    conv("upconv2.conv", nf // 8, nf // 4, 3)
    bn("bn2", nf // 8)
    conv("conv2.0", nf // 8, nf // 8 + feat[0] + 1, 3)
    reduc("reduc2x2", nf // 8, nf // 16)
    conv("upconv1.conv", nf // 16, nf // 8, 3)
    reduc("reduc1x1", nf // 16, nf // 32, True)
    conv("conv1.0", nf // 16, nf // 16 + 4, 3)
    conv("get_depth.0", 1, nf // 16, 3)
    return d


def fill_params(shapes: "OrderedDict[str, Tuple[int, ...]]", seed: int = 0,
                gain: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """Fill a key->shape map from PCG64(seed), iterating keys in SORTED order.

    conv weights: xavier-uniform U(-a, a), a = gain*sqrt(6/(fan_in+fan_out));
    BN weight U(0.5,1.5), bias N(0,0.1), running_mean N(0,0.1), running_var U(0.5,1.5).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    out: Dict[str, np.ndarray] = {}
    for key in sorted(shapes.keys()):
        shp = shapes[key]
        if key.endswith("num_batches_tracked"):
            out[key] = np.asarray(0, dtype=np.int64)
        elif len(shp) == 4:
            cout, cin, kh, kw = shp
            a = gain * math.sqrt(6.0 / (cin * kh * kw + cout * kh * kw))
            out[key] = rng.uniform(-a, a, size=shp).astype(np.float32)
        elif key.endswith("running_var") or key.endswith(".weight"):
            out[key] = rng.uniform(0.5, 1.5, size=shp).astype(np.float32)
        else:  # BN bias / running_mean
            out[key] = (0.1 * rng.standard_normal(size=shp)).astype(np.float32)
    return OrderedDict((k, out[k]) for k in shapes.keys())


def decoder_state(feat: Sequence[int], nf: int = 512, seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    return fill_params(decoder_param_shapes(feat, nf), seed)


def encoder_features(feat: Sequence[int], B: int, H: int, W: int, seed: int = 1234):
    """Encoder-shaped synthetic features: relu(N(0,1)) skips at
    H/2,H/4,H/8,H/16 and N(0,1) at H/32.  Returned list matches the reference
    ``encoder.forward`` output: [x, f1/2, f1/4, f1/8, f1/16, f1/32] (bts.py:327-338);
    element 0 (the image) is not read by the decoder and is None here."""
    assert H % 32 == 0 and W % 32 == 0
    rng = np.random.Generator(np.random.PCG64(seed))
    feats = [None]
    for i, c in enumerate(feat):
        s = 2 ** (i + 1)
        a = rng.standard_normal(size=(B, c, H // s, W // s), dtype=np.float32)
        if i < 4:
            a = np.maximum(a, 0.0)
        feats.append(a)
    return feats


def focal_values(B: int, dataset: str = "kitti", seed: int = 1234) -> np.ndarray:
    if dataset != "kitti":
        return np.full((B,), NYU_FOCAL, dtype=np.float32)
    rng = np.random.Generator(np.random.PCG64(seed + 77))
    idx = rng.integers(0, len(KITTI_FOCALS), size=B)
    return np.asarray([KITTI_FOCALS[i] for i in idx], dtype=np.float32)


def image_batch(B: int, H: int, W: int, seed: int = 1234) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.standard_normal(size=(B, 3, H, W), dtype=np.float32)


def train_targets(B: int, H: int, W: int, max_depth: float, seed: int):
    """Synthetic ground-truth depth [B,1,H,W] in (0.5, max_depth) and a ~70 % validity mask for training-step
    tests and goldens (the reference masks invalid lidar pixels, bts_main.py:478-481)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    gt = rng.uniform(0.5, max_depth, size=(B, 1, H, W)).astype(np.float32)
    mask = rng.uniform(size=(B, 1, H, W)) > 0.3
    return gt, mask
