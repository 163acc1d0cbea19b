"""Drop-in for ``fashion_mnist.DiffusionLayer`` of the reference."""
from ..layers import FashionDiffusionLayer as DiffusionLayer  # noqa: F401

__all__ = ["DiffusionLayer"]
