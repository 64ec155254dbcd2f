"""Encoder backbones for BtsModel, written in plain torch.nn (torchvision is not installed here).

The reference builds its encoder from ``torchvision.models.<arch>(pretrained=True)``
(reference pytorch/bts.py:299-323) and walks ``base_model._modules`` (bts.py:327-338).  These
classes reproduce torchvision's module tree and parameter names so that a reference checkpoint
(``encoder.base_model.*`` keys) loads with ``load_state_dict`` unchanged.  Weights are random
(no network): pass a checkpoint through ``BtsModel.load_state_dict`` as bts_test.py:99-100 does.

The encoder is caller-side of the hot path: it stays on PyTorch-ROCm/MIOpen.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Sequence

import torch
import torch.nn as nn


# --------------------------------------------------------------------------- DenseNet
class _DenseLayer(nn.Module):
    def __init__(self, num_input_features: int, growth_rate: int, bn_size: int):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(num_input_features)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(num_input_features, bn_size * growth_rate, kernel_size=1, stride=1, bias=False)
        self.norm2 = nn.BatchNorm2d(bn_size * growth_rate)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(bn_size * growth_rate, growth_rate, kernel_size=3, stride=1, padding=1, bias=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.conv2(self.relu2(self.norm2(self.conv1(self.relu1(self.norm1(x))))))


class _DenseBlock(nn.ModuleDict):
    """torchvision's _DenseBlock: each layer consumes the concatenation of all earlier features."""

    def __init__(self, num_layers: int, num_input_features: int, bn_size: int, growth_rate: int):
        super().__init__()
        self.num_input_features = num_input_features
        self.growth_rate = growth_rate
        for i in range(num_layers):
            self.add_module("denselayer%d" % (i + 1),
                            _DenseLayer(num_input_features + i * growth_rate, growth_rate, bn_size))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        feats = [x]
        for _, layer in self.items():
            feats.append(layer(torch.cat(feats, 1)))
        return torch.cat(feats, 1)


class _Transition(nn.Sequential):
    def __init__(self, num_input_features: int, num_output_features: int):
        super().__init__()
        self.add_module("norm", nn.BatchNorm2d(num_input_features))
        self.add_module("relu", nn.ReLU(inplace=True))
        self.add_module("conv", nn.Conv2d(num_input_features, num_output_features, kernel_size=1, stride=1, bias=False))
        self.add_module("pool", nn.AvgPool2d(kernel_size=2, stride=2))


def densenet_features(growth_rate: int, block_config: Sequence[int], num_init_features: int,
                      bn_size: int = 4) -> nn.Sequential:
    """``torchvision.models.DenseNet(...).features`` (the reference keeps only ``.features``)."""
    features = nn.Sequential(OrderedDict([
        ("conv0", nn.Conv2d(3, num_init_features, kernel_size=7, stride=2, padding=3, bias=False)),
        ("norm0", nn.BatchNorm2d(num_init_features)),
        ("relu0", nn.ReLU(inplace=True)),
        ("pool0", nn.MaxPool2d(kernel_size=3, stride=2, padding=1)),
    ]))
    nfeat = num_init_features
    for i, num_layers in enumerate(block_config):
        features.add_module("denseblock%d" % (i + 1), _DenseBlock(num_layers, nfeat, bn_size, growth_rate))
        nfeat += num_layers * growth_rate
        if i != len(block_config) - 1:
            features.add_module("transition%d" % (i + 1), _Transition(nfeat, nfeat // 2))
            nfeat //= 2
    features.add_module("norm5", nn.BatchNorm2d(nfeat))
    for m in features.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)
    return features


# ------------------------------------------------------------------------- ResNe(X)t
class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int, downsample, groups: int, base_width: int):
        super().__init__()
        width = int(planes * (base_width / 64.0)) * groups
        self.conv1 = nn.Conv2d(inplanes, width, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, kernel_size=3, stride=stride, padding=1, groups=groups, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, planes * self.expansion, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class ResNet(nn.Module):
    """torchvision.models.ResNet with Bottleneck blocks (resnet50/101, resnext50_32x4d, resnext101_32x8d).
    The reference iterates ``_modules`` and skips 'avgpool'/'fc' (bts.py:330-332); both exist here so
    checkpoint keys match."""

    def __init__(self, layers: Sequence[int], groups: int = 1, width_per_group: int = 64, num_classes: int = 1000):
        super().__init__()
        self.inplanes = 64
        self.groups = groups
        self.base_width = width_per_group
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0], 1)
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * _Bottleneck.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        for m in self.modules():        # zero-init the last BN of each block keeps random-init activations bounded
            if isinstance(m, _Bottleneck):
                nn.init.constant_(m.bn3.weight, 0.2)

    def _make_layer(self, planes: int, blocks: int, stride: int) -> nn.Sequential:
        downsample = None
        if stride != 1 or self.inplanes != planes * _Bottleneck.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * _Bottleneck.expansion, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * _Bottleneck.expansion))
        layers = [_Bottleneck(self.inplanes, planes, stride, downsample, self.groups, self.base_width)]
        self.inplanes = planes * _Bottleneck.expansion
        for _ in range(1, blocks):
            layers.append(_Bottleneck(self.inplanes, planes, 1, None, self.groups, self.base_width))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def build_base_model(name: str) -> nn.Module:
    """``params.encoder`` -> base model, as reference bts.py:300-323 (without pretrained weights)."""
    if name == "densenet121_bts":
        return densenet_features(32, (6, 12, 24, 16), 64)
    if name == "densenet161_bts":
        return densenet_features(48, (6, 12, 36, 24), 96)
    if name == "resnet50_bts":
        return ResNet((3, 4, 6, 3))
    if name == "resnet101_bts":
        return ResNet((3, 4, 23, 3))
    if name == "resnext50_bts":
        return ResNet((3, 4, 6, 3), groups=32, width_per_group=4)
    if name == "resnext101_bts":
        return ResNet((3, 4, 23, 3), groups=32, width_per_group=8)
    raise ValueError("Not supported encoder: {}".format(name))
