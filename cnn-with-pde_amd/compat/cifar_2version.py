"""Drop-ins for ``cifar_2version.LearnableDiffusionLayer`` and the Ruthotto-Haber blocks of the reference."""
from ..layers import LearnableDiffusionLayer as LearnableDiffusionLayer  # noqa: F401
from ..models import SymmetricLayer, ParabolicBlock, HamiltonianBlock  # noqa: F401

__all__ = ["LearnableDiffusionLayer", "SymmetricLayer", "ParabolicBlock", "HamiltonianBlock"]
