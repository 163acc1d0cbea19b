"""Drop-in for ``cifar10.EnhancedDiffusionLayer`` of the reference."""
from ..layers import EnhancedDiffusionLayer as EnhancedDiffusionLayer  # noqa: F401

__all__ = ["EnhancedDiffusionLayer"]
