"""Public surface of cnn_with_pde_amd."""
from . import _lib
from ._lib import PdeError, LIB_PATH
from .functional import (Sweep, adi_schedule, adi_diffuse, adi_diffuse_mixed, adi_diffuse_small, adi_diffuse_multi, gate_combine, bn_pool,
                         plan_checkpoints, channel_mix, explicit5_step, jacobi_diffuse, timing_enable, timing_read)
from .dist import shard_range, shard_batch, GradBucket
from .layers import (MnistDiffusionLayer, FashionDiffusionLayer, SvhnDiffusionLayer, EnhancedDiffusionLayer,
                     LearnableDiffusionLayer, ImprovedDiffusionLayer, PDELayer)
from . import models
from . import graphs
from .graphs import freeze_checkpoint_plans, make_graphed, GraphedStep
from .models import (MnistPDEClassifier, FashionPDEClassifier, SvhnPDEClassifier, SpatialAttention, MultiScaleExtractor,
                     EnhancedFC, CIFAR10PDENoConv, SymmetricLayer, ParabolicBlock, HamiltonianBlock, HybridPDEExtractor,
                     NonConvSpatialAttention, HybridClassifierHead, CIFAR10HybridPDEModel, hybrid_pde_regularization,
                     TinyImageNetClassifier, EmotionDiffusionClassifier, diffuse_shared_input)

#: (reference script, reference class name) -> class here
REFERENCE_CLASSES = {
    ("mnist_test", "DiffusionLayer"): MnistDiffusionLayer,
    ("fashion_mnist", "DiffusionLayer"): FashionDiffusionLayer,
    ("SVHN", "DiffusionLayer"): SvhnDiffusionLayer,
    ("cifar10", "EnhancedDiffusionLayer"): EnhancedDiffusionLayer,
    ("cifar_2version", "LearnableDiffusionLayer"): LearnableDiffusionLayer,
    ("tiny_imagenet", "ImprovedDiffusionLayer"): ImprovedDiffusionLayer,
    ("emotion_recognition", "PDELayer"): PDELayer,
    ("mnist_test", "PDEClassifier"): MnistPDEClassifier,
    ("fashion_mnist", "FashionPDEClassifier"): FashionPDEClassifier,
    ("SVHN", "PDEClassifier"): SvhnPDEClassifier,
    ("cifar10", "SpatialAttention"): SpatialAttention,
    ("cifar10", "MultiScaleExtractor"): MultiScaleExtractor,
    ("cifar10", "EnhancedFC"): EnhancedFC,
    ("cifar10", "CIFAR10PDENoConv"): CIFAR10PDENoConv,
    ("cifar_2version", "SymmetricLayer"): SymmetricLayer,
    ("cifar_2version", "ParabolicBlock"): ParabolicBlock,
    ("cifar_2version", "HamiltonianBlock"): HamiltonianBlock,
    ("cifar_2version", "HybridPDEExtractor"): HybridPDEExtractor,
    ("cifar_2version", "NonConvSpatialAttention"): NonConvSpatialAttention,
    ("cifar_2version", "PDEClassifier"): HybridClassifierHead,
    ("cifar_2version", "CIFAR10HybridPDEModel"): CIFAR10HybridPDEModel,
    ("tiny_imagenet", "ImprovedTinyImageNetClassifier"): TinyImageNetClassifier,
    ("emotion_recognition", "DiffusionClassifier"): EmotionDiffusionClassifier,
}


def library_version() -> str:
    return _lib.load().pde_version().decode()


__all__ = ["graphs", "freeze_checkpoint_plans", "make_graphed", "GraphedStep", "PdeError", "LIB_PATH", "Sweep", "adi_schedule", "adi_diffuse", "adi_diffuse_mixed", "adi_diffuse_small",
           "adi_diffuse_multi", "gate_combine", "bn_pool", "plan_checkpoints", "channel_mix", "explicit5_step",
           "jacobi_diffuse", "timing_enable", "timing_read", "MnistDiffusionLayer", "FashionDiffusionLayer",
           "SvhnDiffusionLayer", "EnhancedDiffusionLayer", "LearnableDiffusionLayer", "ImprovedDiffusionLayer",
           "PDELayer", "models", "MnistPDEClassifier", "FashionPDEClassifier", "SvhnPDEClassifier", "SpatialAttention",
           "MultiScaleExtractor", "EnhancedFC", "CIFAR10PDENoConv", "SymmetricLayer", "ParabolicBlock", "HamiltonianBlock",
           "HybridPDEExtractor", "NonConvSpatialAttention", "HybridClassifierHead", "CIFAR10HybridPDEModel",
           "hybrid_pde_regularization", "TinyImageNetClassifier",
           "EmotionDiffusionClassifier", "diffuse_shared_input", "REFERENCE_CLASSES", "library_version", "shard_range", "shard_batch", "GradBucket"]
