"""Drop-in for ``emotion_recognition.PDELayer`` of the reference."""
from ..layers import PDELayer as PDELayer  # noqa: F401

__all__ = ["PDELayer"]
