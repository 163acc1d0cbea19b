"""nn.Module surface of the reference's PDE layers, running on libpdecnn_hip.so.

Every class keeps the reference's constructor signature (positional order), parameter
names / shapes / initial values (they are state_dict keys, optimizer-group selectors and
regulariser selectors — SURVEY.md §8b), plain attributes and helper methods, so a reference
``state_dict`` loads unchanged and the reference's training / plotting code can keep reaching
into ``model.diff.alpha_base`` etc.  ``forward`` returns a new tensor and never mutates its
input.  There is no CPU path: calling a layer on CPU tensors raises.

    reference class                               here
    mnist_test.DiffusionLayer                     MnistDiffusionLayer
    fashion_mnist.DiffusionLayer                  FashionDiffusionLayer
    SVHN.DiffusionLayer                           SvhnDiffusionLayer
    cifar10.EnhancedDiffusionLayer                EnhancedDiffusionLayer
    cifar_2version.LearnableDiffusionLayer        LearnableDiffusionLayer
    tiny_imagenet.ImprovedDiffusionLayer          ImprovedDiffusionLayer
    emotion_recognition.PDELayer                  PDELayer

``compat/<script>.py`` re-exports each one under the reference's own class name.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as F_
from . import _lib as L

__all__ = ["MnistDiffusionLayer", "FashionDiffusionLayer", "SvhnDiffusionLayer", "EnhancedDiffusionLayer",
           "LearnableDiffusionLayer", "ImprovedDiffusionLayer", "PDELayer"]


class _AdiBase(nn.Module):
    """Shared machinery of the implicit (Thomas) layers."""
    _split = "strang"
    _smooth3 = False
    _clamp_max = None

    def _schedule(self):
        """Per-step sweep lists; cached per (dt, dx, dy, steps) — the attributes may be changed from outside."""
        dy = getattr(self, "dy", self.dx)
        key = (self.dt, self.dx, dy, self.num_steps, self._split)
        cache = self.__dict__.setdefault("_schedule_cache", {})
        steps = cache.get(key)
        if steps is None:
            cache.clear()
            steps = cache[key] = F_.Schedule(F_.adi_schedule(self.dt, self.dx, dy, self.num_steps, self._split))
        return steps

    def get_alpha_beta_at_time(self, t):
        """clamp(base + slope*t) — mnist_test.py:33-42 / cifar10.py:53-63 (host-side helper)."""
        alpha_t = self.alpha_base + self.alpha_time_coeff * t
        beta_t = self.beta_base + self.beta_time_coeff * t
        if self._clamp_max is None:
            return torch.clamp(alpha_t, min=self.stability_eps), torch.clamp(beta_t, min=self.stability_eps)
        return (torch.clamp(alpha_t, min=self.stability_eps, max=self._clamp_max),
                torch.clamp(beta_t, min=self.stability_eps, max=self._clamp_max))

    #: How the backward's checkpoints are chosen (functional.plan_checkpoints).
    #: "auto" (default): from THIS call's coefficients.  Their per-sweep maxima leave the device right
    #: behind the forward's factorisation kernel (pinned copy + event recorded inside pde_adi_forward, before
    #: the sweep launch), so the backward's wait for them is over long before it is reached; exact after
    #: load_state_dict, a change of dt/dx/dy, or any optimiser step.
    #: "lagged": from the coefficients of the previous call in grad mode (never waits; half the error budget
    #: to cover one optimiser step of drift).  The cache is dropped by load_state_dict and by a change of
    #: dt/dx/dy/num_steps, but a LARGE in-place jump of the parameters between two calls is not seen: use it
    #: only in loops whose parameters move by optimiser steps.
    #: An int is an explicit bit mask.
    checkpoint_policy = "auto"

    def _load_from_state_dict(self, *args, **kwargs):
        self.__dict__.pop("_kmax_cache", None)              # a lagged plan made for the old parameters is void
        return super()._load_from_state_dict(*args, **kwargs)

    def _lagged_plan(self, key, u, args, kw, flat, plan):
        """(mask, cache, key): the plan from the previous call's coefficients; the first call (or the first after
        the cache was dropped) computes them synchronously."""
        dy = getattr(self, "dy", self.dx)
        key = key + (self.dt, self.dx, dy, self.num_steps, u.device)
        cache = self.__dict__.setdefault("_kmax_cache", {})
        old = cache.get(key)
        if old is None:
            for k in [k for k in cache if k[-5:] != key[-5:]]:      # plans made for another dt/dx/dy/num_steps/device
                del cache[k]
            km = F_.kappa_max_async(u, *args, flat, **kw)
            km.event.synchronize()
            old = (km, plan(km.host.tolist()))
        elif old[0].event.query():
            old = (old[0], plan(old[0].host.tolist()))
        return old, cache, key

    def _uses_operator(self):
        """Whether the layer runs a channel operator between its steps (then checkpoint masks are step-local)."""
        return False

    def _step_groups(self, steps):
        per = max(1, L.PDE_MAX_SWEEPS // len(steps[0]))
        return [steps[i:i + per] for i in range(0, len(steps), per)]

    def freeze_checkpoint_plan(self, example=None):
        """Pin the checkpoint plan to explicit masks made from the CURRENT parameters alone — one small kernel and one
        synchronous wait per group of steps, no forward pass of the layer (nothing else in the model runs, no BatchNorm
        statistics move, no random numbers are drawn); the conservative budget of the lagged policy.  A call with explicit
        masks issues launches only — no wait for the coefficient maxima, no host copy — which is what hipGraph capture needs
        (``graphs.py``).  The masks stay valid while the coefficients do not grow by more than the budget's margin (a factor
        2 in error amplification); call again after large parameter changes.  ``example`` is accepted for compatibility and
        ignored.  Returns the mask (an int) or, for schedules that run as several launch groups, one mask per group."""
        ab = self.alpha_base
        Cc, N = (1, ab.shape[-1]) if ab.dim() == 2 else (ab.shape[0], ab.shape[-1])
        like = torch.empty((1, Cc, N, N), dtype=torch.float32, device=ab.device)
        args = (self.alpha_base, self.beta_base, self.alpha_time_coeff, self.beta_time_coeff)
        kw = dict(smooth3=self._smooth3, clamp_max=self._clamp_max, eps=self.stability_eps)
        masks = []
        with torch.no_grad():
            for grp in self._step_groups(self._schedule()):
                sps = len(grp[0])
                km = F_.kappa_max_async(like, *args, [s for st in grp for s in st], **kw)
                km.event.synchronize()
                v = km.host.tolist()
                if self._uses_operator():
                    bits = 0
                    for k in range(len(grp)):
                        bits |= F_.plan_checkpoints(v[k * sps:(k + 1) * sps], F_.CKPT_AMAX / 2)
                else:
                    bits = F_.plan_checkpoints(v, F_.CKPT_AMAX / 2)
                masks.append(int(bits))
        self.__dict__.pop("_kmax_cache", None)
        self.checkpoint_policy = masks[0] if len(masks) == 1 else tuple(masks)
        return self.checkpoint_policy

    def _policy_of_group(self, gi, num_sweeps=None):
        """The checkpoint policy of launch group ``gi``: a frozen plan is one mask per group; a single mask given for a
        schedule of several groups keeps the bits the group's own schedule has (a plain group of S sweeps: bits 0..S-2)."""
        ck = self.checkpoint_policy
        if isinstance(ck, (tuple, list)):
            ck = ck[gi]
        if isinstance(ck, int) and not isinstance(ck, bool) and num_sweeps is not None:
            ck &= (1 << max(num_sweeps - 1, 0)) - 1
        return ck

    def _diffuse(self, u, sweeps, gi=0):
        args = (self.alpha_base, self.beta_base, self.alpha_time_coeff, self.beta_time_coeff)
        kw = dict(smooth3=self._smooth3, clamp_max=self._clamp_max, eps=self.stability_eps)
        ck = self._policy_of_group(gi, len(sweeps))
        if not (torch.is_grad_enabled() and (u.requires_grad or any(p.requires_grad for p in args))):
            ck = 0
        if ck != "lagged":
            return F_.adi_diffuse(u, *args, sweeps, checkpoints=ck, **kw)
        old, cache, key = self._lagged_plan(("plain", len(sweeps), sweeps[0].t), u, args, kw, sweeps,
                                            lambda km: F_.plan_checkpoints(km, F_.CKPT_AMAX / 2))
        sink = []
        y = F_.adi_diffuse(u, *args, sweeps, checkpoints=old[1], kmax_sink=sink, **kw)
        cache[key] = (sink[0], old[1]) if sink else old
        return y

    def _run(self, u, steps, M=None, mode=None, skip_weight=None):
        """All steps of the layer.  One launch sequence holds at most PDE_MAX_SWEEPS sweeps: longer schedules
        (num_steps > 32 Strang steps) are cut into groups of whole steps, chained through autograd."""
        per = max(1, L.PDE_MAX_SWEEPS // len(steps[0]))
        if len(steps) <= per:
            if M is None:
                return self._diffuse(u, steps.flat if isinstance(steps, F_.Schedule) else [s for st in steps for s in st])
            return self._diffuse_mixed(u, steps, M, mode, skip_weight)
        u0 = u
        for gi, grp in enumerate(self._step_groups(steps)):
            u = (self._diffuse(u, [s for st in grp for s in st], gi) if M is None
                 else self._diffuse_mixed(u, grp, M, mode, gi=gi))
        return u if skip_weight is None else F_.skip_blend(u0, u, skip_weight)

    #: False forces the per-step launch path (pde_adi_mixed_*) where the single-launch C <= 4 kernels would apply
    small_channel_kernels = True

    def _diffuse_mixed(self, u, steps, M, mode, skip_weight=None, gi=0):
        """All steps of a layer with a channel operator between them; checkpoints as in ``_diffuse`` (one step-local
        mask for every step).  C <= 4 (the reference's own models): the whole time loop — and for SVHN the skip
        blend — in one launch per pass (functional.adi_diffuse_small); otherwise one mixing and one sweep launch per
        step (functional.adi_diffuse_mixed) and the skip blend as its own pass."""
        args = (self.alpha_base, self.beta_base, self.alpha_time_coeff, self.beta_time_coeff)
        kw = dict(smooth3=self._smooth3, clamp_max=self._clamp_max, eps=self.stability_eps)
        small = self.small_channel_kernels and F_.adi_small_supported(u, steps, **kw)
        u_in = u

        def run(ck, sink=None):
            if small:
                return F_.adi_diffuse_small(u, *args, M, steps, mode, skip_weight, checkpoints=ck, kmax_sink=sink, **kw)
            y = F_.adi_diffuse_mixed(u, *args, M, steps, mode, checkpoints=ck, kmax_sink=sink, **kw)
            return y if skip_weight is None else F_.skip_blend(u_in, y, skip_weight)          # SVHN.py:73-74

        ck = self._policy_of_group(gi)
        live = (M, skip_weight) + args
        if not (torch.is_grad_enabled() and (u.requires_grad or any(p is not None and p.requires_grad for p in live))):
            ck = 0
        if ck != "lagged":
            return run(ck)
        sps = len(steps[0])

        def plan(km):
            bits = 0
            for k in range(len(steps)):
                bits |= F_.plan_checkpoints(km[k * sps:(k + 1) * sps], F_.CKPT_AMAX / 2)
            return bits
        old, cache, key = self._lagged_plan(("mixed", len(steps), sps, steps[0][0].t), u, args, kw,
                                            [s for st in steps for s in st], plan)
        sink = []
        y = run(old[1], sink)
        cache[key] = (sink[0], old[1]) if sink else old
        return y


class MnistDiffusionLayer(_AdiBase):
    """mnist_test.py:11-219.  (B,1,size,size) -> same; Strang split, smoothed coefficients."""
    _smooth3 = True

    def __init__(self, size=28, dt=0.001, dx=1.0, dy=1.0, num_steps=10):
        super().__init__()
        self.size, self.dt, self.dx, self.dy, self.num_steps = size, dt, dx, dy, num_steps
        self.alpha_base = nn.Parameter(torch.ones(size, size) * 2.0)
        self.beta_base = nn.Parameter(torch.ones(size, size) * 2.0)
        self.alpha_time_coeff = nn.Parameter(torch.zeros(size, size))
        self.beta_time_coeff = nn.Parameter(torch.zeros(size, size))
        self.stability_eps = 1e-6
        print(f"Initialized DiffusionLayer with dx={dx}, dy={dy}")

    def forward(self, u):
        if u.dim() != 4 or u.shape[1] != 1:
            raise ValueError(f"expected (B,1,{self.size},{self.size}), got {tuple(u.shape)}")
        return self._run(u, self._schedule())

    def get_numerical_stability_info(self):
        """mnist_test.py:200-219."""
        with torch.no_grad():
            alpha_max = torch.max(self.alpha_base + torch.abs(self.alpha_time_coeff) * self.dt * self.num_steps)
            beta_max = torch.max(self.beta_base + torch.abs(self.beta_time_coeff) * self.dt * self.num_steps)
            cfl_x = alpha_max * self.dt / (self.dx ** 2)
            cfl_y = beta_max * self.dt / (self.dy ** 2)
            return {"cfl_x": cfl_x.item(), "cfl_y": cfl_y.item(), "dx": self.dx, "dy": self.dy, "dt": self.dt,
                    "stable_x": cfl_x.item() < 0.5, "stable_y": cfl_y.item() < 0.5}


class FashionDiffusionLayer(_AdiBase):
    """fashion_mnist.py:18-196: the mnist layer with dy == dx, dt=0.3, 4 steps, base 1.8."""
    _smooth3 = True

    def __init__(self, size=28, dt=0.3, dx=1.0, num_steps=4):
        super().__init__()
        self.size, self.dt, self.dx, self.num_steps = size, dt, dx, num_steps
        self.alpha_base = nn.Parameter(torch.ones(size, size) * 1.8)
        self.beta_base = nn.Parameter(torch.ones(size, size) * 1.8)
        self.alpha_time_coeff = nn.Parameter(torch.zeros(size, size))
        self.beta_time_coeff = nn.Parameter(torch.zeros(size, size))
        self.stability_eps = 1e-6

    def forward(self, u):
        if u.dim() != 4 or u.shape[1] != 1:
            raise ValueError(f"expected (B,1,{self.size},{self.size}), got {tuple(u.shape)}")
        return self._run(u, self._schedule())


class SvhnDiffusionLayer(_AdiBase):
    """SVHN.py:12-230: per-channel smoothed Strang steps, channel coupling after every step
    (SVHN.py:71), learnable sigmoid skip blend with the input (SVHN.py:74)."""
    _smooth3 = True

    def __init__(self, size=32, channels=3, dt=0.01, dx=1.0, num_steps=10):
        super().__init__()
        self.size, self.channels, self.dt, self.dx, self.num_steps = size, channels, dt, dx, num_steps
        self.alpha_base = nn.Parameter(torch.ones(channels, size, size) * 0.1)
        self.beta_base = nn.Parameter(torch.ones(channels, size, size) * 0.1)
        self.alpha_time_coeff = nn.Parameter(torch.randn(channels, size, size) * 0.001)
        self.beta_time_coeff = nn.Parameter(torch.randn(channels, size, size) * 0.001)
        self.channel_coupling = nn.Parameter(torch.eye(channels) * 0.01)
        self.stability_eps = 1e-6
        self.skip_weight = nn.Parameter(torch.tensor(0.9))

    def _uses_operator(self):
        return True

    def forward(self, u):
        # SVHN.py:55-76: sweeps, coupling after every step, then sigmoid(w) u0 + (1 - sigmoid(w)) u
        return self._run(u, self._schedule(), self.channel_coupling, "post", self.skip_weight)


class EnhancedDiffusionLayer(_AdiBase):
    """cifar10.py:24-211: coefficients clamped to [eps,10], no smoothing, channel mixing before
    every Strang step (cifar10.py:91).

    ``channel_mixing_enabled=False`` (keyword only, not in the reference) skips the C x C
    mixing product; the whole time loop then runs as ONE fused kernel launch.  It equals the
    reference layer with ``channel_mixing`` set to the identity."""
    _clamp_max = 10.0

    def __init__(self, size=32, channels=3, dt=0.001, dx=1.0, dy=1.0, num_steps=10, *, channel_mixing_enabled=True):
        super().__init__()
        self.size, self.channels, self.dt, self.dx, self.dy, self.num_steps = size, channels, dt, dx, dy, num_steps
        self.alpha_base = nn.Parameter(torch.ones(channels, size, size) * 1.0)
        self.beta_base = nn.Parameter(torch.ones(channels, size, size) * 1.0)
        self.alpha_time_coeff = nn.Parameter(torch.zeros(channels, size, size) * 0.1)
        self.beta_time_coeff = nn.Parameter(torch.zeros(channels, size, size) * 0.1)
        self.channel_mixing = nn.Parameter(torch.eye(channels) + torch.randn(channels, channels) * 0.01)
        self.stability_eps = 1e-6
        self.channel_mixing_enabled = channel_mixing_enabled
        self._banner()

    def _banner(self):
        print(f"Alpha/Beta-Focused DiffusionLayer: {self.size}x{self.size}x{self.channels}")
        print(f"  Spatial: dx={self.dx}, dy={self.dy}")
        print(f"  Temporal: dt={self.dt}, steps={self.num_steps}")
        print(f"  Learnable parameters: α matrices ({self.channels}x{self.size}x{self.size}), "
              f"β matrices ({self.channels}x{self.size}x{self.size})")

    def _uses_operator(self):
        return bool(self.channel_mixing_enabled)

    def forward(self, u):
        steps = self._schedule()
        if not self.channel_mixing_enabled:
            return self._run(u, steps)
        return self._run(u, steps, self.channel_mixing, "pre")


class LearnableDiffusionLayer(EnhancedDiffusionLayer):
    """cifar_2version.py:20-187: as EnhancedDiffusionLayer with the Lie split x(dt/2), y(dt/2)."""
    _split = "lie"

    def _banner(self):
        print(f"Learnable Diffusion Layer: {self.size}x{self.size}x{self.channels}")
        print(f"  Learnable α coefficients: {self.channels}x{self.size}x{self.size}")
        print(f"  Learnable β coefficients: {self.channels}x{self.size}x{self.size}")
        print(f"  Temporal: dt={self.dt}, steps={self.num_steps}")


class ImprovedDiffusionLayer(nn.Module):
    """tiny_imagenet.py:14-72 (the live part): per-channel scaled explicit 5-point step with a
    0.1 relaxation.  ``beta_base`` exists but is unused, as in the reference (its grad is None).
    The reference's training script reads ``model.diff.spatial_modulation`` (tiny_imagenet.py:614),
    an attribute its own class never defines; it is not defined here either."""

    def __init__(self, size=64, channels=3, dt=0.01, num_steps=1, use_implicit=False):
        super().__init__()
        self.size, self.channels, self.dt, self.num_steps, self.use_implicit = size, channels, dt, num_steps, use_implicit
        self.alpha_base = nn.Parameter(torch.ones(channels) * 0.05)
        self.beta_base = nn.Parameter(torch.ones(channels) * 0.05)
        self.channel_scaling = nn.Parameter(torch.ones(channels))
        self.stability_eps = 1e-6
        self.max_coeff = 0.15

    def forward(self, u):
        # all num_steps steps in one call (64x64 / 32x32 / 16x16 planes: one launch, the plane stays in registers)
        return F_.explicit5_step(u, self.alpha_base, self.channel_scaling, self.dt, self.stability_eps,
                                 self.max_coeff, 0.1, self.num_steps)


class PDELayer(nn.Module):
    """emotion_recognition.py:56-97: reflect-pad once, Nt explicit Jacobi updates with separable
    coefficients alpha(y) (rows) and beta(x) (columns) made from six scalars."""

    def __init__(self, Nx=48, Ny=48, Lx=1.0, Ly=1.0, T=0.01, dt=0.001):
        super().__init__()
        self.Nx, self.Ny, self.Lx, self.Ly = Nx, Ny, Lx, Ly
        self.T, self.dt = T, dt
        self.dx = Lx / Nx
        self.dy = Ly / Ny
        self.Nt = int(T / dt)
        self.alpha_w1 = nn.Parameter(torch.tensor(0.1))
        self.alpha_w2 = nn.Parameter(torch.tensor(0.1))
        self.alpha_w3 = nn.Parameter(torch.tensor(0.1))
        self.beta_w1 = nn.Parameter(torch.tensor(0.3))
        self.beta_w2 = nn.Parameter(torch.tensor(0.2))
        self.beta_w3 = nn.Parameter(torch.tensor(0.2))
        self.register_buffer("x", torch.linspace(0, Lx, Nx))
        self.register_buffer("y", torch.linspace(0, Ly, Ny))

    def alpha(self, y_val):
        return 0.5 * self.dt * (self.alpha_w1 + self.alpha_w2 * torch.sin(2 * torch.pi * y_val)
                                + self.alpha_w3 * torch.sin(4 * torch.pi * y_val)) / self.dx ** 2

    def beta(self, x_val):
        return self.dt * (self.beta_w1 + self.beta_w2 * torch.cos(2 * torch.pi * x_val)
                          + self.beta_w3 * torch.cos(4 * torch.pi * x_val)) / self.dy ** 2

    def forward(self, u0):
        if u0.dim() != 4 or u0.shape[1] != 1:
            raise ValueError(f"expected (B,1,H,W), got {tuple(u0.shape)}")
        # coefficient vectors: alpha varies along dim 1 (rows), beta along dim 2 (columns)
        out = F_.jacobi_diffuse(u0.squeeze(1), self.alpha(self.y), self.beta(self.x), self.Nt)
        return out.unsqueeze(1)
