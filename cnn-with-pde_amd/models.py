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
    cifar_2version.SymmetricLayer / ParabolicBlock / HamiltonianBlock :190-258   same names (fp32-MFMA kernels, pde_rh.hip)
    cifar_2version.HybridPDEExtractor   :261-302           HybridPDEExtractor    (two diffusion layers in ONE launch per pass)
    cifar_2version.NonConvSpatialAttention / PDEClassifier / CIFAR10HybridPDEModel :305-412
                                                           NonConvSpatialAttention / HybridClassifierHead / CIFAR10HybridPDEModel
    cifar_2version.hybrid_pde_regularization :415-436      hybrid_pde_regularization
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
           "EnhancedFC", "CIFAR10PDENoConv", "SymmetricLayer", "ParabolicBlock", "HamiltonianBlock", "HybridPDEExtractor",
           "NonConvSpatialAttention", "HybridClassifierHead", "CIFAR10HybridPDEModel", "hybrid_pde_regularization",
           "TinyImageNetClassifier", "EmotionDiffusionClassifier", "diffuse_shared_input"]


def diffuse_shared_input(layers, x, weights=None, plane_sums=False):
    """Run mixing-first PDE layers (``EnhancedDiffusionLayer`` / ``LearnableDiffusionLayer``) that share the input
    ``x`` in ONE launch per pass when the library supports the shapes (C <= 4), else one after the other.
    Returns ``(sum_i weights[i] * y_i or None, [y_1 .. y_L])``; with ``plane_sums`` also ``[sum_hw y_i]`` (B,C)."""
    steps = [ly._schedule() for ly in layers]
    same_split = len({len(st[0]) for st in steps}) == 1
    fused = (same_split and len(layers) <= 4 and x.is_cuda and all(getattr(ly, "channel_mixing_enabled", True) for ly in layers)
             and all(ly.small_channel_kernels and ly.checkpoint_policy != "lagged" for ly in layers)
             and (all(ly.checkpoint_policy == "auto" for ly in layers)
                  or all(isinstance(ly.checkpoint_policy, int) for ly in layers))          # frozen masks may differ per layer
             and all(F_.adi_small_supported(x, st, smooth3=ly._smooth3, clamp_max=ly._clamp_max, eps=ly.stability_eps)
                     for ly, st in zip(layers, steps)))
    if not fused:
        ys = [ly(x) for ly in layers]
        out = None if weights is None else sum(w * y for w, y in zip(weights, ys))
        return (out, ys, [y.sum(dim=(2, 3)) for y in ys]) if plane_sums else (out, ys)
    descr = [dict(alpha_base=ly.alpha_base, beta_base=ly.beta_base, alpha_time_coeff=ly.alpha_time_coeff,
                  beta_time_coeff=ly.beta_time_coeff, M=ly.channel_mixing, steps=st, smooth3=ly._smooth3,
                  clamp_max=ly._clamp_max, eps=ly.stability_eps) for ly, st in zip(layers, steps)]
    pol = layers[0].checkpoint_policy
    res = F_.adi_diffuse_multi(x, descr, weights, plane_sums, pol if pol == "auto" else tuple(int(ly.checkpoint_policy) for ly in layers))
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

    #: BatchNorm2d + the two 4x4 poolings in two passes over `combined` (functional.bn_pool) instead of torch's five
    fused_tail = True

    def forward(self, x):
        combined = self.feature_extractor(x)[0]
        if self.fused_tail and F_.bn_pool_supported(combined, self.feature_bn):
            pooled = F_.bn_pool(combined, self.feature_bn)                    # cifar10.py:346-353
        else:
            feats = self.feature_bn(combined)
            pooled = torch.cat([self.adaptive_pool(feats), self.max_pool(feats)], dim=1)
        return self.classifier(pooled.view(pooled.size(0), -1))


class SymmetricLayer(nn.Module):
    """cifar_2version.py:190-220 (Ruthotto & Haber): F(Y) = -K^T act(BN(K Y)) on the flattened image, K a dense
    (C*H*W)^2 matrix initialised near the identity.  On the GPU both products, the BatchNorm1d over the batch and the
    activation run on the fp32 matrix cores (``functional.sym_layer``, pde_rh.hip — SURVEY §8f-4: up to 128 rows each product
    as 32-column strips with the contraction split over workgroups plus a small epilogue launch);
    ``residual(base, X, scale)`` is the fused update ``base + scale * (act(BN(X K^T)) K)`` the two blocks below are made
    of.  ``fused = False`` (or a shape the kernels do not take, or autocast) is plain torch, as in the reference."""

    def __init__(self, channels, spatial_size, activation="relu"):
        super().__init__()
        self.channels, self.spatial_size = channels, spatial_size
        self.feature_dim = channels * spatial_size * spatial_size
        self.K = nn.Linear(self.feature_dim, self.feature_dim, bias=False)
        self.norm = nn.BatchNorm1d(self.feature_dim)
        self.act_name = activation if activation in ("relu", "tanh") else "identity"
        self.activation = {"relu": nn.ReLU, "tanh": nn.Tanh}.get(activation, nn.Identity)()
        self.fused = True
        nn.init.eye_(self.K.weight)
        self.K.weight.data += torch.randn_like(self.K.weight) * 0.01

    def _fused_ok(self, X):
        return self.fused and F_.sym_layer_supported(X, self.norm)

    def residual(self, base, X, scale):
        """base + scale * (act(BN(X K^T)) K): one step of the Parabolic / Hamiltonian blocks."""
        if self._fused_ok(X):
            return F_.sym_layer(X, self.K.weight, self.norm, self.act_name, base=base, scale=scale)
        return base + scale * (-self.forward(X))

    def forward(self, Y):
        if self._fused_ok(Y):
            return F_.sym_layer(Y, self.K.weight, self.norm, self.act_name, base=None, scale=-1.0)
        B = Y.shape[0]
        h = self.activation(self.norm(self.K(Y.reshape(B, -1))))
        return (-(h @ self.K.weight)).view_as(Y)


class ParabolicBlock(nn.Module):
    """cifar_2version.py:223-236: forward Euler on dY/dt = F_sym(Y)."""

    def __init__(self, channels, spatial_size, num_steps=3, dt=1.0):
        super().__init__()
        self.num_steps, self.dt = num_steps, dt
        self.symmetric_layer = SymmetricLayer(channels, spatial_size)
        print(f"Parabolic Block: {num_steps} steps, dt={dt}")

    def forward(self, Y):
        for _ in range(self.num_steps):
            Y = self.symmetric_layer.residual(Y, Y, -self.dt)          # Y + dt * F_sym(Y)
        return Y


class HamiltonianBlock(nn.Module):
    """cifar_2version.py:239-258: symplectic steps Y += dt*(-F_Y(Z)); Z -= dt*F_Z(Y), Z starting at zero."""

    def __init__(self, channels, spatial_size, num_steps=3, dt=1.0):
        super().__init__()
        self.num_steps, self.dt = num_steps, dt
        self.F_Y = SymmetricLayer(channels, spatial_size)
        self.F_Z = SymmetricLayer(channels, spatial_size)
        print(f"Hamiltonian Block: {num_steps} steps, dt={dt}")

    def forward(self, Y):
        Z = torch.zeros_like(Y)
        for _ in range(self.num_steps):
            Y = self.F_Y.residual(Y, Z, self.dt)                       # Y - dt * F_Y(Z)
            Z = self.F_Z.residual(Z, Y, self.dt)                       # Z - dt * F_Z(Y)
        return Y


class HybridPDEExtractor(nn.Module):
    """cifar_2version.py:261-302: two LearnableDiffusionLayers (one launch per pass), a Parabolic and a Hamiltonian
    block on the same input, softmax-weighted sum, BatchNorm2d."""

    def __init__(self, input_size=32, channels=3):
        super().__init__()
        self.diffusion1 = LearnableDiffusionLayer(input_size, channels, dt=0.001, num_steps=8)
        self.diffusion2 = LearnableDiffusionLayer(input_size, channels, dt=0.002, num_steps=5)
        self.parabolic = ParabolicBlock(channels, input_size, num_steps=4, dt=0.5)
        self.hamiltonian = HamiltonianBlock(channels, input_size, num_steps=3, dt=0.8)
        self.combination_weights = nn.Parameter(torch.ones(4) / 4)
        self.feature_norm = nn.BatchNorm2d(channels)

    def forward(self, x):
        w = F.softmax(self.combination_weights, dim=0)
        _, (d1, d2) = diffuse_shared_input([self.diffusion1, self.diffusion2], x)
        par, ham = self.parabolic(x), self.hamiltonian(x)
        combined = self.feature_norm(w[0] * d1 + w[1] * d2 + w[2] * par + w[3] * ham)
        return combined, d1, d2, par, ham


class NonConvSpatialAttention(nn.Module):
    """cifar_2version.py:305-328: an element-wise gate from a three-layer MLP of the flattened x + pos_embed."""

    def __init__(self, channels, spatial_size):
        super().__init__()
        self.channels, self.spatial_size = channels, spatial_size
        self.feature_dim = d = channels * spatial_size * spatial_size
        self.pos_embed = nn.Parameter(torch.randn(1, channels, spatial_size, spatial_size) * 0.02)
        self.attention_net = nn.Sequential(nn.Linear(d, d // 4), nn.ReLU(), nn.Linear(d // 4, d // 8), nn.ReLU(),
                                           nn.Linear(d // 8, d), nn.Sigmoid())

    def forward(self, x):
        return x * self.attention_net((x + self.pos_embed).reshape(x.shape[0], -1)).view_as(x)


class HybridClassifierHead(nn.Module):
    """cifar_2version.PDEClassifier :331-366: 1024-512-256-128 Linear / BatchNorm1d / ReLU / Dropout stack."""

    def __init__(self, input_dim, num_classes=10, dropout_rate=0.4):
        super().__init__()
        mods, prev = [], input_dim
        for i, h in enumerate((1024, 512, 256, 128)):
            mods += [nn.Linear(prev, h), nn.BatchNorm1d(h), nn.ReLU(inplace=True),
                     nn.Dropout(dropout_rate // 2 if i == 3 else dropout_rate)]
            prev = h
        mods.append(nn.Linear(prev, num_classes))
        self.classifier = nn.Sequential(*mods)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, x):
        return self.classifier(x)


class CIFAR10HybridPDEModel(nn.Module):
    """cifar_2version.py:369-412: HybridPDEExtractor, element-wise attention, BatchNorm2d, 8x8 average and max pooling,
    classifier on 384 features."""

    def __init__(self, dropout_rate=0.4):
        super().__init__()
        self.feature_extractor = HybridPDEExtractor(input_size=32, channels=3)
        self.attention = NonConvSpatialAttention(channels=3, spatial_size=32)
        self.adaptive_avg_pool = nn.AdaptiveAvgPool2d((8, 8))
        self.adaptive_max_pool = nn.AdaptiveMaxPool2d((8, 8))
        self.feature_bn = nn.BatchNorm2d(3)
        self.classifier = HybridClassifierHead(input_dim=384, num_classes=10, dropout_rate=dropout_rate)

    def forward(self, x):
        feats = self.feature_bn(self.attention(self.feature_extractor(x)[0]))
        pooled = torch.cat([self.adaptive_avg_pool(feats), self.adaptive_max_pool(feats)], dim=1)
        return self.classifier(pooled.view(pooled.size(0), -1))


def hybrid_pde_regularization(model, alpha1=1e-4, alpha2=1e-4, alpha3=1e-6):
    """cifar_2version.py:415-436: penalties selected by parameter NAME (which is why the layers keep the
    reference's names): squared L2 on alpha_base / beta_base and on K.weight, squared Frobenius distance of
    channel_mixing from the identity, L1 on combination_weights."""
    reg = 0.0
    for name, p in model.named_parameters():
        if "alpha_base" in name or "beta_base" in name:
            reg = reg + alpha3 * p.pow(2).sum()
        elif "channel_mixing" in name:
            reg = reg + alpha2 * (p - torch.eye(p.size(0), device=p.device)).pow(2).sum()
        elif "K.weight" in name:
            reg = reg + alpha2 * p.pow(2).sum()
        elif "combination_weights" in name:
            reg = reg + alpha1 * p.abs().sum()
    return reg


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
