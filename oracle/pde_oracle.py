"""CPU oracle for the PDE-layer hot path.  TEST INFRASTRUCTURE — NOT A PRODUCT PATH.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product (``cnn_with_pde_amd``) never does; it
fails loudly when the HIP library is missing.

What this is: a single parametrised restatement, in plain PyTorch-CPU ops with
autograd, of the arithmetic of the reference's seven layer classes:

    K1 (implicit ADI, Thomas line solves)
        mnist_test.DiffusionLayer            mnist_test.py:11-198
        fashion_mnist.DiffusionLayer         fashion_mnist.py:18-196
        SVHN.DiffusionLayer                  SVHN.py:12-230
        cifar10.EnhancedDiffusionLayer       cifar10.py:24-211
        cifar_2version.LearnableDiffusionLayer  cifar_2version.py:20-187
    K2 (explicit 5-point)
        tiny_imagenet.ImprovedDiffusionLayer tiny_imagenet.py:14-72
        emotion_recognition.PDELayer         emotion_recognition.py:56-97
    Ruthotto-Haber blocks (dense symmetric layer)
        cifar_2version.SymmetricLayer / ParabolicBlock / HamiltonianBlock   cifar_2version.py:190-258

Pinning: the reference has no tests or golden vectors of its own,
so the pins are the vectors in ``tests/golden/*.npz`` produced by running the
reference classes themselves in the build container (``tools/make_golden.py``).
``tests/test_oracle_golden.py`` holds this oracle to those vectors (forward,
input gradient and every parameter gradient; bitwise in fp32 for K1).

The time loop, the split, the coefficient construction and the Thomas
recurrences follow the op order of the reference so that fp32 results agree to
the last bit; the loop over the unknown index is a Python loop of vectorised
torch ops exactly as in the reference ("reference-faithful mode" of
BASELINE.md §3), which is what makes this the honest CPU baseline as well.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

EPS = 1e-6


# --------------------------------------------------------------------------- #
# K1: implicit, dimensionally split diffusion                                  #
# --------------------------------------------------------------------------- #
@dataclass(frozen=True)
class AdiSpec:
    """Static description of one K1 layer variant."""
    size: int = 28
    channels: int = 1
    dt: float = 1e-3
    dx: float = 1.0
    dy: float = 1.0
    num_steps: int = 10
    split: str = "strang"            # "strang": x(dt/2) y(dt) x(dt/2); "lie": x(dt/2) y(dt/2)
    smooth3: bool = False            # 3-tap replicate-padded average of the coefficient along the solve axis
    clamp_max: Optional[float] = None
    mix: str = "none"                # "none" | "pre" (cifar: M u before each step) | "post" (SVHN: K u after each step)
    skip: bool = False               # SVHN: sigmoid(s) u0 + (1 - sigmoid(s)) u_S
    eps: float = EPS


def mnist_spec(size=28, dt=0.001, dx=1.0, dy=1.0, num_steps=10) -> AdiSpec:
    """mnist_test.py:12 — single channel, smoothed, Strang, clamp min only."""
    return AdiSpec(size, 1, dt, dx, dy, num_steps, "strang", True, None, "none", False)


def fashion_spec(size=28, dt=0.3, dx=1.0, num_steps=4) -> AdiSpec:
    """fashion_mnist.py:19 — the mnist layer with dy == dx (fashion_mnist.py:63)."""
    return AdiSpec(size, 1, dt, dx, dx, num_steps, "strang", True, None, "none", False)


def svhn_spec(size=32, channels=3, dt=0.01, dx=1.0, num_steps=10) -> AdiSpec:
    """SVHN.py:13 — per-channel mnist layer, post coupling (SVHN.py:71) and skip blend (SVHN.py:74)."""
    return AdiSpec(size, channels, dt, dx, dx, num_steps, "strang", True, None, "post", True)


def cifar10_spec(size=32, channels=3, dt=0.001, dx=1.0, dy=1.0, num_steps=10) -> AdiSpec:
    """cifar10.py:25 — clamp to [eps, 10], no smoothing, channel mixing before every step (cifar10.py:91)."""
    return AdiSpec(size, channels, dt, dx, dy, num_steps, "strang", False, 10.0, "pre", False)


def cifar2_spec(size=32, channels=3, dt=0.001, dx=1.0, dy=1.0, num_steps=10) -> AdiSpec:
    """cifar_2version.py:25 — as cifar10 but Lie split x(dt/2), y(dt/2) (cifar_2version.py:93,99)."""
    return AdiSpec(size, channels, dt, dx, dy, num_steps, "lie", False, 10.0, "pre", False)


def sweep_schedule(spec: AdiSpec) -> List[Tuple[int, float, float]]:
    """[(axis, delta, t)] for every implicit sweep, in execution order.

    axis 0 = "x" (solve along W with alpha), 1 = "y" (solve along H with beta).
    ``t`` accumulates ``dt/2`` in Python double exactly as the reference does
    (mnist_test.py:49-63; cifar_2version.py:79-101).
    """
    out = []
    t = 0.0
    for _ in range(spec.num_steps):
        out.append((0, spec.dt / 2, t))
        t += spec.dt / 2
        if spec.split == "strang":
            out.append((1, spec.dt, t))
            t += spec.dt / 2
            out.append((0, spec.dt / 2, t))
        elif spec.split == "lie":
            out.append((1, spec.dt / 2, t))
            t += spec.dt / 2
        else:
            raise ValueError(spec.split)
    return out


def coefficient_at(base: torch.Tensor, slope: torch.Tensor, t: float, spec: AdiSpec) -> torch.Tensor:
    """clamp(base + slope*t, eps[, max]) — mnist_test.py:33-42; cifar10.py:53-63."""
    theta = base + slope * t
    if spec.clamp_max is None:
        return torch.clamp(theta, min=spec.eps)
    return torch.clamp(theta, min=spec.eps, max=spec.clamp_max)


def _smooth3(lines: torch.Tensor) -> torch.Tensor:
    """3-tap moving average along the last axis with replicate ends (mnist_test.py:135-149)."""
    padded = F.pad(lines, (1, 1), mode="replicate")
    k = torch.ones(1, 1, 3, dtype=lines.dtype) / 3
    return F.conv1d(padded.unsqueeze(1), k, padding=0).squeeze(1)


def thomas_lines(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor, d: torch.Tensor, eps: float) -> torch.Tensor:
    """Solve the (lines, N) tridiagonal systems with the reference's recurrences
    (mnist_test.py:151-198 == cifar10.py:179-211):

        c*_0 = c_0/(b_0+eps)            d*_0 = d_0/(b_0+eps)
        den_i = b_i - a_i c*_{i-1} + eps
        c*_i = c_i/den_i (i<N-1)        d*_i = (d_i - a_i d*_{i-1})/den_i
        x_{N-1} = d*_{N-1}              x_i = d*_i - c*_i x_{i+1}
    """
    n = d.shape[1]
    cs: List[torch.Tensor] = []
    ds: List[torch.Tensor] = []
    den = b[:, 0] + eps
    cs.append(c[:, 0] / den)
    ds.append(d[:, 0] / den)
    for i in range(1, n):
        den = b[:, i] - a[:, i] * cs[i - 1] + eps
        cs.append(c[:, i] / den if i < n - 1 else None)
        ds.append((d[:, i] - a[:, i] * ds[i - 1]) / den)
    xs: List[Optional[torch.Tensor]] = [None] * n
    xs[n - 1] = ds[n - 1]
    for i in range(n - 2, -1, -1):
        xs[i] = ds[i] - cs[i] * xs[i + 1]
    return torch.stack(xs, dim=1)


def _implicit_sweep(u: torch.Tensor, theta: torch.Tensor, axis: int, delta: float, h: float,
                    spec: AdiSpec) -> torch.Tensor:
    """One backward-Euler sweep of (B,C,H,W) along ``axis`` (0: W, 1: H).

    x: mnist_test.py:67-98 / cifar10.py:124-148; y: mnist_test.py:100-133 / cifar10.py:150-177.
    ``theta`` is the clamped (C,H,W) coefficient of this sweep.
    """
    B, C, H, W = u.shape
    if axis == 1:
        u_l = u.transpose(2, 3).contiguous()
        th = theta.transpose(1, 2).contiguous()
    else:
        u_l = u.contiguous()
        th = theta.contiguous()
    n = u_l.shape[3]
    d = u_l.view(B * C * u_l.shape[2], n)
    th_l = th.unsqueeze(0).expand(B, -1, -1, -1).contiguous().view(B * C * th.shape[1], n)
    if spec.smooth3:
        th_l = _smooth3(th_l)
    coeff = th_l * delta / (h ** 2)
    a = -coeff
    c = -coeff
    b = 1 + 2 * coeff
    b = b.clone()
    b[:, 0] = 1 + coeff[:, 0]
    b[:, -1] = 1 + coeff[:, -1]
    x = thomas_lines(a, b, c, d, spec.eps).view(B, C, u_l.shape[2], n)
    if axis == 1:
        x = x.transpose(2, 3).contiguous()
    return x


def adi_forward(u: torch.Tensor, params: Dict[str, torch.Tensor], spec: AdiSpec, state_cast=None) -> torch.Tensor:
    """Forward of a K1 layer.  ``u`` is (B,C,H,W); parameters are (C,H,W)
    (``(H,W)`` is accepted for the single-channel classes) plus the optional
    ``channel_mixing`` / ``channel_coupling`` (C,C) and scalar ``skip_weight``.
    Differentiable through torch autograd, any float dtype.

    ``state_cast`` (tests of reduced-precision tensor I/O only; not part of the reference): a function applied
    to the state after each channel operator, after the sweeps of each time step and to the result — the points
    at which an implementation with bf16 tensors and fp32 arithmetic rounds, e.g.
    ``lambda t: t.bfloat16().float()`` (autograd then rounds the gradient at the same points).
    """
    cast = state_cast if state_cast is not None else (lambda t: t)
    B, C, H, W = u.shape

    def chw(p):
        return p if p.dim() == 3 else p.unsqueeze(0)

    ab, bb = chw(params["alpha_base"]), chw(params["beta_base"])
    asl, bsl = chw(params["alpha_time_coeff"]), chw(params["beta_time_coeff"])
    u0 = u
    sched = sweep_schedule(spec)
    per_step = 3 if spec.split == "strang" else 2
    for k in range(spec.num_steps):
        if spec.mix == "pre":
            # cifar10.py:65-72  out[b,i,p] = sum_j M[i,j] u[b,j,p]
            u = cast(torch.matmul(params["channel_mixing"], u.reshape(B, C, H * W)).view(B, C, H, W))
        for axis, delta, t in sched[k * per_step:(k + 1) * per_step]:
            if axis == 0:
                u = _implicit_sweep(u, coefficient_at(ab, asl, t, spec), 0, delta, spec.dx, spec)
            else:
                u = _implicit_sweep(u, coefficient_at(bb, bsl, t, spec), 1, delta, spec.dy, spec)
        if spec.mix != "none":
            u = cast(u)
        if spec.mix == "post":
            # SVHN.py:78-86  (B*H*W, C) @ K^T
            flat = u.permute(0, 2, 3, 1).contiguous().view(B * H * W, C)
            u = cast(torch.matmul(flat, params["channel_coupling"].t()).view(B, H, W, C).permute(0, 3, 1, 2).contiguous())
    if spec.skip:
        s = torch.sigmoid(params["skip_weight"])
        u = s * u0 + (1 - s) * u                      # SVHN.py:74
    return cast(u)


def adi_init_params(spec: AdiSpec, variant: str, dtype=torch.float32, gen: Optional[torch.Generator] = None
                    ) -> Dict[str, torch.Tensor]:
    """Reference initial values."""
    C, N = spec.channels, spec.size
    shape = (N, N) if variant in ("mnist", "fashion") else (C, N, N)
    base = {"mnist": 2.0, "fashion": 1.8, "svhn": 0.1, "cifar10": 1.0, "cifar2": 1.0}[variant]
    p = {
        "alpha_base": torch.full(shape, base, dtype=dtype),
        "beta_base": torch.full(shape, base, dtype=dtype),
        "alpha_time_coeff": torch.zeros(shape, dtype=dtype),
        "beta_time_coeff": torch.zeros(shape, dtype=dtype),
    }
    if variant == "svhn":
        p["alpha_time_coeff"] = torch.randn(shape, generator=gen, dtype=dtype) * 0.001
        p["beta_time_coeff"] = torch.randn(shape, generator=gen, dtype=dtype) * 0.001
        p["channel_coupling"] = torch.eye(C, dtype=dtype) * 0.01
        p["skip_weight"] = torch.tensor(0.9, dtype=dtype)
    if variant in ("cifar10", "cifar2"):
        p["channel_mixing"] = torch.eye(C, dtype=dtype) + torch.randn(C, C, generator=gen, dtype=dtype) * 0.01
    return p


# --------------------------------------------------------------------------- #
# K2: explicit 5-point layers                                                  #
# --------------------------------------------------------------------------- #
def tiny_forward(u: torch.Tensor, params: Dict[str, torch.Tensor], dt: float = 0.01, num_steps: int = 1,
                 eps: float = EPS, max_coeff: float = 0.15) -> torch.Tensor:
    """tiny_imagenet.py:34-72: v = s_c u;  u <- u + 0.1 (v + a_c dt Lap0(v) - u), zero ghost cells."""
    C = u.shape[1]
    for _ in range(num_steps):
        a = torch.clamp(params["alpha_base"], min=eps, max=max_coeff)
        v = u * params["channel_scaling"].view(1, -1, 1, 1)
        vp = F.pad(v, (1, 1, 1, 1))
        lap = vp[:, :, :-2, 1:-1] + vp[:, :, 2:, 1:-1] + vp[:, :, 1:-1, :-2] + vp[:, :, 1:-1, 2:] - 4 * v
        new = v + (a * dt).view(1, C, 1, 1) * lap
        u = u + 0.1 * (new - u)
    return u


def tiny_forward_faithful(u, params, dt=0.01, num_steps=1, eps=EPS, max_coeff=0.15):
    """Same arithmetic routed through conv2d per channel as the reference does
    (tiny_imagenet.py:59-70); used to pin bitwise and as the CPU baseline."""
    C = u.shape[1]
    k = torch.tensor([[0, 1, 0], [1, -4, 1], [0, 1, 0]], dtype=u.dtype).view(1, 1, 3, 3)
    for _ in range(num_steps):
        a = torch.clamp(params["alpha_base"], min=eps, max=max_coeff)
        v = u * params["channel_scaling"].view(1, -1, 1, 1)
        cols = []
        for c in range(C):
            lap = F.conv2d(v[:, c:c + 1], k, padding=1)
            cols.append(v[:, c:c + 1] + a[c] * dt * lap)
        new = torch.cat(cols, dim=1)
        u = u + 0.1 * (new - u)
    return u


def emotion_coefficients(params: Dict[str, torch.Tensor], Nx=48, Ny=48, Lx=1.0, Ly=1.0, dt=0.001,
                         dtype=torch.float32) -> Tuple[torch.Tensor, torch.Tensor]:
    """A_i = alpha(y_i) (rows), B_j = beta(x_j) (columns) — emotion_recognition.py:76-80,87-89."""
    dx, dy = Lx / Nx, Ly / Ny
    x = torch.linspace(0, Lx, Nx, dtype=dtype)
    y = torch.linspace(0, Ly, Ny, dtype=dtype)
    A = 0.5 * dt * (params["alpha_w1"] + params["alpha_w2"] * torch.sin(2 * torch.pi * y)
                    + params["alpha_w3"] * torch.sin(4 * torch.pi * y)) / dx ** 2
    Bc = dt * (params["beta_w1"] + params["beta_w2"] * torch.cos(2 * torch.pi * x)
               + params["beta_w3"] * torch.cos(4 * torch.pi * x)) / dy ** 2
    return A, Bc


def jacobi_forward(u: torch.Tensor, A: torch.Tensor, Bc: torch.Tensor, nt: int) -> torch.Tensor:
    """The time loop of emotion_recognition.py:85-97 for given coefficient vectors: u (B,H,W),
    A (H,) multiplies the second difference along dim 1, Bc (W,) the one along dim 2."""
    P = F.pad(u, (1, 1, 1, 1), mode="reflect")
    A = A.view(1, -1, 1)
    Bc = Bc.view(1, 1, -1)
    for _ in range(nt):
        inner = P[:, 1:-1, 1:-1]
        d1 = P[:, 2:, 1:-1] - 2 * inner + P[:, :-2, 1:-1]
        d2 = P[:, 1:-1, 2:] - 2 * inner + P[:, 1:-1, :-2]
        new = inner + A * d1 + Bc * d2
        P = torch.cat([P[:, :1], torch.cat([P[:, 1:-1, :1], new, P[:, 1:-1, -1:]], dim=2), P[:, -1:]], dim=1)
    return P[:, 1:-1, 1:-1]


def emotion_forward(u0: torch.Tensor, params: Dict[str, torch.Tensor], Nx=48, Ny=48, Lx=1.0, Ly=1.0,
                    T=0.01, dt=0.001) -> torch.Tensor:
    """emotion_recognition.py:82-97: reflect-pad once, Nt Jacobi updates of the
    interior, the padded ring keeps its initial values."""
    A, Bc = emotion_coefficients(params, Nx, Ny, Lx, Ly, dt, u0.dtype)
    return jacobi_forward(u0.squeeze(1), A, Bc, int(T / dt)).unsqueeze(1)


def emotion_init_params(dtype=torch.float32) -> Dict[str, torch.Tensor]:
    vals = dict(alpha_w1=0.1, alpha_w2=0.1, alpha_w3=0.1, beta_w1=0.3, beta_w2=0.2, beta_w3=0.2)
    return {k: torch.tensor(v, dtype=dtype) for k, v in vals.items()}


# --------------------------------------------------------------------------- #
# helpers shared by tests and bench                                            #
# --------------------------------------------------------------------------- #
# ---- Ruthotto-Haber blocks (cifar_2version.py:190-258) ---------------------------------------------------------
def batch_norm_1d(P: torch.Tensor, weight, bias, running_mean, running_var, training: bool, momentum=0.1, eps=1e-5):
    """nn.BatchNorm1d on a (B, D) input, spelled out (cifar_2version.py:201, 215): batch statistics with the biased
    variance in training mode — the running statistics move by ``momentum`` towards the batch mean and the UNBIASED
    variance —, running statistics otherwise.  Returns (output, new_running_mean, new_running_var)."""
    if training:
        B = P.shape[0]
        mean = P.mean(dim=0)
        var = ((P - mean) ** 2).mean(dim=0)
        new_rm, new_rv = running_mean, running_var
        if running_mean is not None:
            unb = var * (B / (B - 1)) if B > 1 else var
            new_rm = (1 - momentum) * running_mean + momentum * mean.detach()
            new_rv = (1 - momentum) * running_var + momentum * unb.detach()
    else:
        mean, var = running_mean, running_var
        new_rm, new_rv = running_mean, running_var
    out = (P - mean) / torch.sqrt(var + eps) * weight + bias
    return out, new_rm, new_rv


def _rh_act(x, activation: str):
    return torch.relu(x) if activation == "relu" else (torch.tanh(x) if activation == "tanh" else x)      # :203-208


def symmetric_layer(Y: torch.Tensor, sl: Dict[str, torch.Tensor], training: bool, activation: str = "relu"):
    """cifar_2version.py:210-219: F_sym(Y) = -act(BN(Y_flat K^T)) K.  ``sl``: K.weight, norm.weight, norm.bias,
    norm.running_mean, norm.running_var (the last two are replaced by their updated values in training mode)."""
    B = Y.shape[0]
    KY = Y.reshape(B, -1) @ sl["K.weight"].t()                                           # :213
    KYn, rm, rv = batch_norm_1d(KY, sl["norm.weight"], sl["norm.bias"], sl.get("norm.running_mean"),
                                sl.get("norm.running_var"), training)                   # :214
    if rm is not None:
        sl["norm.running_mean"], sl["norm.running_var"] = rm, rv
    return (-(_rh_act(KYn, activation) @ sl["K.weight"])).view_as(Y)                    # :215-219


def _sub(params: Dict[str, torch.Tensor], prefix: str) -> Dict[str, torch.Tensor]:
    return {k[len(prefix):]: v for k, v in params.items() if k.startswith(prefix)}


def parabolic_block(Y, params: Dict[str, torch.Tensor], num_steps: int, dt: float, training: bool):
    """cifar_2version.py:231-236: Y <- Y + dt * F_sym(Y), num_steps times (one SymmetricLayer, reused)."""
    sl = _sub(params, "symmetric_layer.")
    for _ in range(num_steps):
        Y = Y + dt * symmetric_layer(Y, sl, training)
    for k, v in sl.items():
        params["symmetric_layer." + k] = v
    return Y


def hamiltonian_block(Y, params: Dict[str, torch.Tensor], num_steps: int, dt: float, training: bool):
    """cifar_2version.py:249-258: Z = 0; Y <- Y + dt * (-F_Y(Z)); Z <- Z - dt * F_Z(Y)."""
    fy, fz = _sub(params, "F_Y."), _sub(params, "F_Z.")
    Z = torch.zeros_like(Y)
    for _ in range(num_steps):
        Y = Y + dt * (-symmetric_layer(Z, fy, training))
        Z = Z - dt * symmetric_layer(Y, fz, training)
    for k, v in fy.items():
        params["F_Y." + k] = v
    for k, v in fz.items():
        params["F_Z." + k] = v
    return Y


def value_and_grads(fn, u: torch.Tensor, params: Dict[str, torch.Tensor], gy: torch.Tensor):
    """Run ``fn(u, params)`` with autograd; return (y, grad_u, {name: grad})."""
    u = u.detach().clone().requires_grad_(True)
    p = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    y = fn(u, p)
    names = [k for k in p]
    grads = torch.autograd.grad(y, [u] + [p[k] for k in names], gy, allow_unused=True)
    return y.detach(), grads[0], {k: g for k, g in zip(names, grads[1:])}
