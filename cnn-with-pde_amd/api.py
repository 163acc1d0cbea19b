"""Public surface of cnn_with_pde_amd."""
from . import _lib
from ._lib import PdeError, LIB_PATH
from .functional import (Sweep, adi_schedule, adi_diffuse, plan_checkpoints, channel_mix, explicit5_step, jacobi_diffuse,
                         timing_enable, timing_read)
from .dist import shard_range, shard_batch, GradBucket
from .layers import (MnistDiffusionLayer, FashionDiffusionLayer, SvhnDiffusionLayer, EnhancedDiffusionLayer,
                     LearnableDiffusionLayer, ImprovedDiffusionLayer, PDELayer)

#: (reference script, reference class name) -> class here
REFERENCE_CLASSES = {
    ("mnist_test", "DiffusionLayer"): MnistDiffusionLayer,
    ("fashion_mnist", "DiffusionLayer"): FashionDiffusionLayer,
    ("SVHN", "DiffusionLayer"): SvhnDiffusionLayer,
    ("cifar10", "EnhancedDiffusionLayer"): EnhancedDiffusionLayer,
    ("cifar_2version", "LearnableDiffusionLayer"): LearnableDiffusionLayer,
    ("tiny_imagenet", "ImprovedDiffusionLayer"): ImprovedDiffusionLayer,
    ("emotion_recognition", "PDELayer"): PDELayer,
}


def library_version() -> str:
    return _lib.load().pde_version().decode()


__all__ = ["PdeError", "LIB_PATH", "Sweep", "adi_schedule", "adi_diffuse", "plan_checkpoints", "channel_mix", "explicit5_step",
           "jacobi_diffuse", "timing_enable", "timing_read", "MnistDiffusionLayer", "FashionDiffusionLayer",
           "SvhnDiffusionLayer", "EnhancedDiffusionLayer", "LearnableDiffusionLayer", "ImprovedDiffusionLayer",
           "PDELayer", "REFERENCE_CLASSES", "library_version", "shard_range", "shard_batch", "GradBucket"]
