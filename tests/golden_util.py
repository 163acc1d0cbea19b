"""Load tests/golden/*.npz (vectors produced by the reference itself, see
tools/make_golden.py) and map each onto the oracle's / the product's arguments."""
import glob
import json
import os

import numpy as np
import torch

from oracle import pde_oracle as O

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODEL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_models")   # the reference's MODELS around the layers


def names(prefixes=None, f64=None, directory=None):
    out = []
    for p in sorted(glob.glob(os.path.join(directory or GOLDEN_DIR, "*.npz"))):
        n = os.path.basename(p)[:-4]
        if prefixes and not any(n.startswith(x) for x in prefixes):
            continue
        if f64 is not None and n.endswith("_f64") != f64:
            continue
        out.append(n)
    return out


class Golden:
    def __init__(self, name, directory=None):
        z = np.load(os.path.join(directory or GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        self.name = name
        self.meta = json.loads(bytes(z["meta"]).decode())
        self.dtype = getattr(torch, self.meta["dtype"])
        t = lambda a: torch.from_numpy(np.array(a))
        self.u, self.gy, self.y, self.gu = t(z["u"]), t(z["gy"]), t(z["y"]), t(z["gu"])
        self.params = {k[6:]: t(z[k]) for k in z.files if k.startswith("param_")}
        self.grads = {k[5:]: t(z[k]) for k in z.files if k.startswith("grad_")}
        self.grad_is_none = {k[9:]: bool(z[k]) for k in z.files if k.startswith("gradnone_")}
        # model fixtures that hold buffers: before the forward, and what one forward left in them (training mode)
        self.bufin = {k[6:]: t(z[k]) for k in z.files if k.startswith("bufin_")}
        self.bufout = {k[7:]: t(z[k]) for k in z.files if k.startswith("bufout_")}
        self.script, self.cls, self.ctor = self.meta["script"], self.meta["cls"], self.meta["ctor"]

    # -- mapping onto the oracle -------------------------------------------------
    def family(self):
        return {"mnist_test": "adi", "fashion_mnist": "adi", "SVHN": "adi", "cifar10": "adi",
                "cifar_2version": "adi", "tiny_imagenet": "tiny", "emotion_recognition": "emotion"}[self.script]

    def adi_spec(self):
        mk = {"mnist_test": O.mnist_spec, "fashion_mnist": O.fashion_spec, "SVHN": O.svhn_spec,
              "cifar10": O.cifar10_spec, "cifar_2version": O.cifar2_spec}[self.script]
        return mk(**self.ctor)

    def oracle_fn(self):
        fam = self.family()
        if fam == "adi":
            spec = self.adi_spec()
            return lambda u, p: O.adi_forward(u, p, spec)
        if fam == "tiny":
            kw = {k: v for k, v in self.ctor.items() if k in ("dt", "num_steps")}
            return lambda u, p: O.tiny_forward(u, p, **kw)
        kw = dict(self.ctor)
        return lambda u, p: O.emotion_forward(u, p, **kw)


def rel_err(a, b):
    """max|a-b| / max|b| — the parity metric used throughout (tolerances are
    stated against this number in each test)."""
    a, b = a.double(), b.double()
    den = float(b.abs().max())
    if den == 0.0:
        return float((a - b).abs().max())
    return float((a - b).abs().max()) / den
