"""Import name for the package whose sources live in ``../cnn-with-pde_amd/``.

The directory name the project mandates contains a hyphen and cannot be imported;
this shim only extends ``__path__`` so that ``cnn_with_pde_amd.<module>`` resolves to
``cnn-with-pde_amd/<module>.py``.
"""
import os as _os

_src = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "cnn-with-pde_amd")
__path__.append(_src)

from .api import *  # noqa: E402,F401,F403
from .api import __all__  # noqa: E402,F401
