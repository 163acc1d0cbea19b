"""Drop-in for ``mnist_test.DiffusionLayer`` of the reference."""
from ..layers import MnistDiffusionLayer as DiffusionLayer  # noqa: F401

__all__ = ["DiffusionLayer"]
