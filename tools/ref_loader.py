"""Load the reference's layer classes from /root/reference (this container only).

TEST INFRASTRUCTURE.  Used by tools/make_golden.py to generate tests/golden/*.npz
and by tests that pin the oracle against the live reference when it is present.
The reference never travels to the GPU box; everything here refuses to run when
/root/reference is absent.

The reference scripts import torchvision / kagglehub / seaborn at module import
(mnist_test.py:7, emotion_recognition.py:11,14); those are absent here and are
irrelevant to the layer classes, so empty placeholder modules are registered
before loading.  Every training entry point is behind ``if __name__ ==
"__main__"`` so nothing is downloaded or trained at import.
"""
import contextlib
import importlib.util
import io
import os
import sys
import types

REF_ROOT = "/root/reference"

_PLACEHOLDERS = [
    "torchvision", "torchvision.datasets", "torchvision.transforms",
    "torchvision.models", "kagglehub", "seaborn",
]


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "mnist_test.py"))


def _install_placeholders():
    for name in _PLACEHOLDERS:
        if name in sys.modules:
            continue
        try:
            importlib.import_module(name)
            continue
        except Exception:
            pass
        mod = types.ModuleType(name)
        mod.__path__ = []  # behave like a package
        sys.modules[name] = mod
        if "." in name:
            parent, child = name.rsplit(".", 1)
            setattr(sys.modules[parent], child, mod)


_cache = {}


def load(script: str):
    """Return the reference script ``script`` (e.g. "mnist_test") as a module."""
    if not reference_available():
        raise RuntimeError("reference tree not present (expected only in the build container)")
    if script in _cache:
        return _cache[script]
    import matplotlib
    matplotlib.use("Agg")
    _install_placeholders()
    path = os.path.join(REF_ROOT, script + ".py")
    spec = importlib.util.spec_from_file_location("_pderef_" + script, path)
    mod = importlib.util.module_from_spec(spec)
    sys.dont_write_bytecode = True
    with contextlib.redirect_stdout(io.StringIO()):
        spec.loader.exec_module(mod)
    _cache[script] = mod
    return mod


@contextlib.contextmanager
def quiet():
    """Silence the constructor banners (mnist_test.py:31, cifar10.py:48-51)."""
    with contextlib.redirect_stdout(io.StringIO()):
        yield
