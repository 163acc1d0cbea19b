"""Counterparts of the reference's models around the PDE layers, so that the layers have the
callers they have in the reference and a reference ``state_dict`` loads unchanged (same attribute names).

Only the PDE layers run on the library; the heads are stock ``torch.nn`` (Linear / BatchNorm / Dropout / Conv2d),
as they are in the reference (out of scope as code to accelerate).

    reference                                              here
    mnist_test.PDEClassifier            :223-237           MnistPDEClassifier
    fashion_mnist.FashionPDEClassifier  :200-224           FashionPDEClassifier
    SVHN.PDEClassifier                  :234-270           SvhnPDEClassifier
    cifar10.SpatialAttention            :215-244           SpatialAttention
    cifar10.MultiScaleExtractor         :248-282           MultiScaleExtractor   (three layers, ONE launch per pass)
    cifar10.EnhancedFC                  :286-314           EnhancedFC
    cifar10.CIFAR10PDENoConv            :318-361           CIFAR10PDENoConv
    cifar_2version.HybridPDEExtractor   :261-302           DiffusionPair         (its two diffusion branches only)
    tiny_imagenet.ImprovedTinyImageNetClassifier :237-329  TinyImageNetClassifier
    emotion_recognition.DiffusionClassifier :170-195       EmotionDiffusionClassifier
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as F_
from .layers import (MnistDiffusionLayer, FashionDiffusionLayer, SvhnDiffusionLayer, EnhancedDiffusionLayer,
                     LearnableDiffusionLayer, ImprovedDiffusionLayer, PDELayer)

__all__ = ["MnistPDEClassifier", "FashionPDEClassifier", "SvhnPDEClassifier", "SpatialAttention", "MultiScaleExtractor",
           "EnhancedFC", "CIFAR10PDENoConv", "DiffusionPair", "TinyImageNetClassifier", "EmotionDiffusionClassifier",
           "diffuse_shared_input"]


def diffuse_shared_input(layers, x, weights=None, plane_sums=False):
    """Run mixing-first PDE layers (``EnhancedDiffusionLayer`` / ``LearnableDiffusionLayer``) that share the input
    ``x`` in ONE launch per pass when the library supports the shapes (C <= 4), else one after the other.
    Returns ``(sum_i weights[i] * y_i or None, [y_1 .. y_L])``; with ``plane_sums`` also ``[sum_hw y_i]`` (B,C)."""
    steps = [ly._schedule() for ly in layers]
    same_split = len({len(st[0]) for st in steps}) == 1
    fused = (same_split and len(layers) <= 4 and x.is_cuda and all(getattr(ly, "channel_mixing_enabled", True) for ly in layers)
             and all(ly.small_channel_kernels and ly.checkpoint_policy == "auto" for ly in layers)
             and all(F_.adi_small_supported(x, st, smooth3=ly._smooth3, clamp_max=ly._clamp_max, eps=ly.stability_eps)
                     for ly, st in zip(layers, steps)))
    if not fused:
        ys = [ly(x) for ly in layers]
        out = None if weights is None else sum(w * y for w, y in zip(weights, ys))
        return (out, ys, [y.sum(dim=(2, 3)) for y in ys]) if plane_sums else (out, ys)
    descr = [dict(alpha_base=ly.alpha_base, beta_base=ly.beta_base, alpha_time_coeff=ly.alpha_time_coeff,
                  beta_time_coeff=ly.beta_time_coeff, M=ly.channel_mixing, steps=st, smooth3=ly._smooth3,
                  clamp_max=ly._clamp_max, eps=ly.stability_eps) for ly, st in zip(layers, steps)]
    res = F_.adi_diffuse_multi(x, descr, weights, plane_sums)
    out = res[0] if weights is not None else None
    return (out, res[1], res[2]) if plane_sums else (out, res[1])


class MnistPDEClassifier(nn.Module):
    """mnist_test.py:223-237: PDE layer, then 784 -> 256 -> 10 with dropout."""

    def __init__(self, dropout_rate=0.1, dx=1.0, dy=1.0):
        super().__init__()
        self.diff = MnistDiffusionLayer(dx=dx, dy=dy)
        self.dropout = nn.Dropout(dropout_rate)
        self.fc1 = nn.Linear(28 * 28, 256)
        self.fc2 = nn.Linear(256, 10)

    def forward(self, x):
        x = self.diff(x).reshape(x.size(0), -1)
        x = self.dropout(F.relu(self.fc1(self.dropout(x))))
        return self.fc2(x)


class FashionPDEClassifier(nn.Module):
    """fashion_mnist.py:200-224: PDE layer, then 784 -> 512 -> 256 -> 10 with batch norm and dropout."""

    def __init__(self, dropout_rate=0.15):
        super().__init__()
        self.diff = FashionDiffusionLayer()
        self.dropout = nn.Dropout(dropout_rate)
        self.fc1 = nn.Linear(28 * 28, 512)
        self.fc2 = nn.Linear(512, 256)
        self.fc3 = nn.Linear(256, 10)
        self.bn1 = nn.BatchNorm1d(512)
        self.bn2 = nn.BatchNorm1d(256)

    def forward(self, x):
        x = self.diff(x).reshape(x.size(0), -1)
        x = self.dropout(F.relu(self.bn1(self.fc1(x))))
        x = self.dropout(F.relu(self.bn2(self.fc2(x))))
        return self.fc3(x)


class SvhnPDEClassifier(nn.Module):
    """SVHN.py:234-270: PDE layer (C = 3, skip blend), then 3072 -> 2048 -> 1024 -> 512 -> 256 -> 10."""

    def __init__(self, dropout_rate=0.5):
        super().__init__()
        self.diff = SvhnDiffusionLayer(size=32, channels=3)
        self.dropout = nn.Dropout(dropout_rate)
        widths = [32 * 32 * 3, 2048, 1024, 512, 256]
        for i in range(4):
            setattr(self, f"fc{i + 1}", nn.Linear(widths[i], widths[i + 1]))
            setattr(self, f"bn{i + 1}", nn.BatchNorm1d(widths[i + 1]))
        self.fc5 = nn.Linear(256, 10)

    def forward(self, x):
        x = self.diff(x).reshape(x.size(0), -1)
        for i in range(1, 5):
            x = self.dropout(F.relu(getattr(self, f"bn{i}")(getattr(self, f"fc{i}")(x))))
        return self.fc5(x)


class SpatialAttention(nn.Module):
    """cifar10.py:215-244: a per-(sample, channel) gate from the spatial mean of ``x + pos_embed``."""

    def __init__(self, channels, size):
        super().__init__()
        self.channels, self.size = channels, size
        self.pos_embed = nn.Parameter(torch.randn(1, channels, size, size) * 0.1)
        self.attention_fc = nn.Sequential(nn.Linear(channels, channels * 2), nn.ReLU(),
                                          nn.Linear(channels * 2, channels), nn.Sigmoid())

    def gate(self, plane_sums, hw):
        """The gate from the spatial SUM of x (B,C): mean(x + pos_embed) = sum(x)/HW + mean(pos_embed)."""
        return self.attention_fc(plane_sums / hw + self.pos_embed.mean(dim=(2, 3)))

    def forward(self, x):
        B, C = x.shape[:2]
        pooled = F.adaptive_avg_pool2d(x + self.pos_embed, (1, 1)).view(B, C)
        return x * self.attention_fc(pooled).view(B, C, 1, 1)


class MultiScaleExtractor(nn.Module):
    """cifar10.py:248-282: three EnhancedDiffusionLayers with different (dt, steps, dx) on the same input, an
    attention gate on each, softmax-weighted sum.  The three PDE layers run in ONE launch forward and ONE backward
    (functional.adi_diffuse_multi) instead of three launch sequences."""

    def __init__(self, input_size=32, channels=3):
        super().__init__()
        self.pde1 = EnhancedDiffusionLayer(input_size, channels, dt=0.001, num_steps=5, dx=1.0, dy=1.0)
        self.pde2 = EnhancedDiffusionLayer(input_size, channels, dt=0.002, num_steps=8, dx=2.0, dy=2.0)
        self.pde3 = EnhancedDiffusionLayer(input_size, channels, dt=0.005, num_steps=4, dx=1.5, dy=1.5)
        self.attention1 = SpatialAttention(channels, input_size)
        self.attention2 = SpatialAttention(channels, input_size)
        self.attention3 = SpatialAttention(channels, input_size)
        self.combine_weights = nn.Parameter(torch.ones(3) / 3)
        print("Multi-scale α/β learning: 3 PDE layers with different temporal/spatial scales")

    #: False: return (combined, None, None, None) and never materialise the three gated feature maps (the reference's
    #: own model, cifar10.py:343, drops them)
    return_features = True
    #: False: attention and combination as plain torch ops (x + pos, pool, multiply, weighted adds: ~14 passes)
    fused_epilogue = True

    def forward(self, x):
        att = (self.attention1, self.attention2, self.attention3)
        w = F.softmax(self.combine_weights, dim=0)
        if not (self.fused_epilogue and x.is_cuda):
            _, ys = diffuse_shared_input([self.pde1, self.pde2, self.pde3], x)
            f1, f2, f3 = (a(y) for a, y in zip(att, ys))
            return w[0] * f1 + w[1] * f2 + w[2] * f3, f1, f2, f3
        # the average pool comes out of the PDE kernel, the (B,C) MLPs stay in torch, gate multiply and weighted sum
        # are one pass (functional.gate_combine): SURVEY §8f-3
        _, ys, sums = diffuse_shared_input([self.pde1, self.pde2, self.pde3], x, plane_sums=True)
        hw = x.shape[2] * x.shape[3]
        gates = [a.gate(s, hw) for a, s in zip(att, sums)]
        combined = F_.gate_combine(ys, gates, w)
        if not self.return_features:
            return combined, None, None, None
        B, C = x.shape[:2]
        return (combined, *[y * g.view(B, C, 1, 1) for y, g in zip(ys, gates)])


class EnhancedFC(nn.Module):
    """cifar10.py:286-314: Linear / BatchNorm1d / ReLU / Dropout stack, Kaiming-normal weights."""

    def __init__(self, input_size, hidden_sizes, num_classes, dropout_rate=0.3):
        super().__init__()
        mods, prev = [], input_size
        for h in hidden_sizes:
            mods += [nn.Linear(prev, h), nn.BatchNorm1d(h), nn.ReLU(inplace=True), nn.Dropout(dropout_rate)]
            prev = h
        mods.append(nn.Linear(prev, num_classes))
        self.network = nn.Sequential(*mods)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.kaiming_normal_(m.weight)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        return self.network(x)


class CIFAR10PDENoConv(nn.Module):
    """cifar10.py:318-361: MultiScaleExtractor, BatchNorm2d, 4x4 average and max pooling, EnhancedFC on 96 features."""

    def __init__(self, dropout_rate=0.3):
        super().__init__()
        self.feature_extractor = MultiScaleExtractor(input_size=32, channels=3)
        self.feature_extractor.return_features = False       # forward() below uses `combined` only (cifar10.py:343)
        self.adaptive_pool = nn.AdaptiveAvgPool2d((4, 4))
        self.max_pool = nn.AdaptiveMaxPool2d((4, 4))
        self.classifier = EnhancedFC(96, [512, 256, 128, 64], 10, dropout_rate)
        self.feature_bn = nn.BatchNorm2d(3)

    def forward(self, x):
        combined = self.feature_extractor(x)[0]
        feats = self.feature_bn(combined)
        pooled = torch.cat([self.adaptive_pool(feats), self.max_pool(feats)], dim=1)
        return self.classifier(pooled.view(pooled.size(0), -1))


class DiffusionPair(nn.Module):
    """The two diffusion branches of cifar_2version.HybridPDEExtractor (:269-270, :287-288): LearnableDiffusionLayers
    with (dt, steps) = (0.001, 8) and (0.002, 5) on the same input, one launch per pass.  (The Parabolic and
    Hamiltonian branches of that extractor are dense 3072 x 3072 layers: SURVEY.md §8f-4, not built.)"""

    def __init__(self, input_size=32, channels=3):
        super().__init__()
        self.diffusion1 = LearnableDiffusionLayer(input_size, channels, dt=0.001, num_steps=8)
        self.diffusion2 = LearnableDiffusionLayer(input_size, channels, dt=0.002, num_steps=5)
        self.combination_weights = nn.Parameter(torch.ones(2) / 2)

    def forward(self, x):
        w = F.softmax(self.combination_weights, dim=0)
        combined, (y1, y2) = diffuse_shared_input([self.diffusion1, self.diffusion2], x, w)
        return combined, y1, y2


class _BasicBlock(nn.Module):
    """tiny_imagenet.py:308-329."""

    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.shortcut = nn.Sequential()
        if stride != 1 or in_planes != planes:
            self.shortcut = nn.Sequential(nn.Conv2d(in_planes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return F.relu(out + self.shortcut(x))


class TinyImageNetClassifier(nn.Module):
    """tiny_imagenet.py:237-305: explicit PDE layer on the image, then a ResNet-18-like CNN."""

    def __init__(self, num_classes=200, use_pde=True, dropout_rate=0.3):
        super().__init__()
        self.use_pde = use_pde
        if use_pde:
            self.diff = ImprovedDiffusionLayer(size=64, channels=3, num_steps=1, use_implicit=False)
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cfg, prev = [(64, 1), (128, 2), (256, 2), (512, 2)], 64
        for i, (planes, stride) in enumerate(cfg):
            setattr(self, f"layer{i + 1}", nn.Sequential(_BasicBlock(prev, planes, stride), _BasicBlock(planes, planes, 1)))
            prev = planes
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.dropout = nn.Dropout(dropout_rate)
        self.fc = nn.Linear(512, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        if self.use_pde:
            x = self.diff(x)
        x = self.maxpool(F.relu(self.bn1(self.conv1(x))))
        for i in range(1, 5):
            x = getattr(self, f"layer{i}")(x)
        return self.fc(self.dropout(torch.flatten(self.avgpool(x), 1)))


class EmotionDiffusionClassifier(nn.Module):
    """emotion_recognition.py:170-195: PDELayer on a (B,1,48,48) image, then 2304 -> 512 -> 256 -> 128 -> 7 with
    batch norm and dropout."""

    def __init__(self, img_size=48, num_classes=7, dropout_rate=0.3):
        super().__init__()
        self.pde = PDELayer(Nx=img_size, Ny=img_size)
        mods, prev = [nn.Flatten()], img_size * img_size
        for h in (512, 256, 128):
            mods += [nn.Linear(prev, h), nn.BatchNorm1d(h), nn.ReLU(), nn.Dropout(dropout_rate)]
            prev = h
        mods.append(nn.Linear(prev, num_classes))
        self.classifier = nn.Sequential(*mods)

    def forward(self, x):
        return self.classifier(self.pde(x))
