"""Drop-in for ``tiny_imagenet.ImprovedDiffusionLayer`` of the reference."""
from ..layers import ImprovedDiffusionLayer as ImprovedDiffusionLayer  # noqa: F401

__all__ = ["ImprovedDiffusionLayer"]
