"""Drop-in for ``cifar_2version.LearnableDiffusionLayer`` of the reference."""
from ..layers import LearnableDiffusionLayer as LearnableDiffusionLayer  # noqa: F401

__all__ = ["LearnableDiffusionLayer"]
