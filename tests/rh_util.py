"""Fixtures of the Ruthotto-Haber blocks (tests/golden_models/model_rh_*.npz, made by the reference's own modules through
tools/make_golden.py) mapped onto the oracle's and the product's arguments."""
import json
import os

import numpy as np
import torch

from oracle import pde_oracle as O

MODEL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_models")
NAMES = [f"model_rh_{k}_8_{m}" for k in ("symmetric", "symmetric_tanh", "parabolic", "hamiltonian") for m in ("train", "eval")]


class RhGolden:
    def __init__(self, name):
        z = np.load(os.path.join(MODEL_DIR, name + ".npz"), allow_pickle=False)
        t = lambda a: torch.from_numpy(np.array(a))
        self.name = name
        self.meta = json.loads(bytes(z["meta"]).decode())
        self.cls, self.ctor = self.meta["cls"], self.meta["ctor"]
        self.training = name.endswith("_train")
        self.u, self.gy, self.y, self.gu = t(z["u"]), t(z["gy"]), t(z["y"]), t(z["gu"])
        self.params = {k[6:]: t(z[k]) for k in z.files if k.startswith("param_")}
        self.grads = {k[5:]: t(z[k]) for k in z.files if k.startswith("grad_")}
        self.bufin = {k[6:]: t(z[k]) for k in z.files if k.startswith("bufin_")}
        self.bufout = {k[7:]: t(z[k]) for k in z.files if k.startswith("bufout_")}
        self.extra = {k: t(z[k]) for k in z.files if k in ("v1", "v2", "gK_v1", "v2_gK", "gK_absmax")}

    def oracle_fn(self):
        """(u, params-with-buffers) -> y; params is updated in place with the new running statistics"""
        c = self.ctor
        if self.cls == "SymmetricLayer":
            return lambda u, p: O.symmetric_layer(u, p, self.training, c.get("activation", "relu"))
        if self.cls == "ParabolicBlock":
            return lambda u, p: O.parabolic_block(u, p, c["num_steps"], c["dt"], self.training)
        return lambda u, p: O.hamiltonian_block(u, p, c["num_steps"], c["dt"], self.training)


def oracle_run(fn, u, params, buffers, gy):
    """y, gu, {param: grad}, {buffer: value after the forward} of the oracle function"""
    u = u.detach().clone().requires_grad_(True)
    p = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    allp = dict(p)
    allp.update({k: v.detach().clone() for k, v in buffers.items() if v.dtype.is_floating_point})
    y = fn(u, allp)
    names = list(p)
    gs = torch.autograd.grad(y, [u] + [p[n] for n in names], gy)
    return y.detach(), gs[0], dict(zip(names, gs[1:])), {k: allp[k].detach() for k in buffers if k in allp}
