"""Drop-in for ``SVHN.DiffusionLayer`` of the reference."""
from ..layers import SvhnDiffusionLayer as DiffusionLayer  # noqa: F401

__all__ = ["DiffusionLayer"]
