"""Differentiable operators of the PDE-layer hot path, backed by libpdecnn_hip.so.

Each function is a ``torch.autograd.Function`` whose forward and backward enqueue the
hand-written HIP kernels on the current stream.  PyTorch is used only for device
memory, streams and autograd bookkeeping.  Nothing here has a CPU or eager fallback.
"""
from __future__ import annotations

import ctypes as C
import threading
import time
import weakref
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib as L

__all__ = ["Sweep", "adi_schedule", "adi_diffuse", "adi_diffuse_mixed", "adi_diffuse_small", "adi_small_supported", "adi_diffuse_multi", "gate_combine", "plan_checkpoints", "kappa_max_async", "channel_mix", "skip_blend", "explicit5_step", "jacobi_diffuse",
           "timing_enable", "timing_read", "Schedule", "sym_layer", "sym_layer_supported"]


@dataclass(frozen=True)
class Sweep:
    """One implicit sweep: a diffuse_x / diffuse_y call of the reference (mnist_test.py:55-63)."""
    axis: int        # 0: along W with alpha, 1: along H with beta
    delta: float     # dt/2 or dt
    h2: float        # dx**2 or dy**2
    t: float         # current_time at which alpha/beta are evaluated


class _HashedTuple(tuple):
    """A tuple whose hash is computed once: schedules are dictionary keys on every layer call (launch descriptors,
    support checks), and hashing 30 frozen dataclasses costs more host time than a kernel launch."""

    def __hash__(self):
        h = self.__dict__.get("_h")
        if h is None:
            h = self.__dict__["_h"] = tuple.__hash__(self)
        return h

    def __eq__(self, other):
        return self is other or tuple.__eq__(self, other)

    def __ne__(self, other):
        return not self.__eq__(other)


class Schedule(_HashedTuple):
    """Per-step sweep tuples of one layer call; ``.flat`` is the same sweeps as one flat tuple."""

    def __new__(cls, steps):
        self = super().__new__(cls, (_HashedTuple(st) for st in steps))
        self.flat = _HashedTuple(s for st in self for s in st)
        return self


def _as_schedule(steps) -> "Schedule":
    return steps if isinstance(steps, Schedule) else Schedule(steps)


def adi_schedule(dt: float, dx: float, dy: float, num_steps: int, split: str = "strang") -> List[List[Sweep]]:
    """Sweeps of every time step, with ``current_time`` accumulated in Python double exactly
    as the reference does (mnist_test.py:49-63 Strang; cifar_2version.py:79-101 Lie)."""
    steps, t = [], 0.0
    for _ in range(num_steps):
        s = [Sweep(0, dt / 2, dx ** 2, t)]
        t += dt / 2
        if split == "strang":
            s.append(Sweep(1, dt, dy ** 2, t))
            t += dt / 2
            s.append(Sweep(0, dt / 2, dx ** 2, t))
        elif split == "lie":
            s.append(Sweep(1, dt / 2, dy ** 2, t))
            t += dt / 2
        else:
            raise ValueError(f"unknown split {split!r}")
        steps.append(s)
    return steps


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.PdeError("libpdecnn_hip operators need CUDA/HIP tensors (there is no CPU fallback)")


def _io_dtype(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return L.PDE_IO_F32
    if t.dtype == torch.bfloat16:
        return L.PDE_IO_BF16
    raise L.PdeError(f"unsupported tensor dtype {t.dtype} (float32 or bfloat16)")


_desc_cache = {}
_M64 = (1 << 64) - 1


def _make_desc(B, Cc, N, io, sweeps: Sequence[Sweep], smooth3, clamp_max, eps) -> L.PdeAdiDesc:
    """The launch descriptor (read-only for the library).  Cached: filling ~100 ctypes fields costs more
    host time than the kernels of a small layer take on the device."""
    if not isinstance(sweeps, tuple):
        sweeps = tuple(sweeps)
    key = (B, Cc, N, io, sweeps, bool(smooth3), clamp_max, float(eps))
    d = _desc_cache.get(key)
    if d is None:
        if len(_desc_cache) > 256:
            _desc_cache.clear()
        d = _desc_cache[key] = _build_desc(B, Cc, N, io, sweeps, smooth3, clamp_max, eps)
    return d


def _build_desc(B, Cc, N, io, sweeps: Sequence[Sweep], smooth3, clamp_max, eps) -> L.PdeAdiDesc:
    if len(sweeps) > L.PDE_MAX_SWEEPS:
        raise L.PdeError(f"{len(sweeps)} sweeps in one launch exceed PDE_MAX_SWEEPS={L.PDE_MAX_SWEEPS}")
    d = L.PdeAdiDesc()
    d.B, d.C, d.N, d.io_dtype, d.num_sweeps = B, Cc, N, io, len(sweeps)
    d.smooth3 = int(bool(smooth3))
    d.has_clamp_max = int(clamp_max is not None)
    d.clamp_max = float(clamp_max) if clamp_max is not None else 0.0
    d.eps = float(eps)
    for i, s in enumerate(sweeps):
        d.sweep[i].axis, d.sweep[i].delta, d.sweep[i].h2, d.sweep[i].t = int(s.axis), s.delta, s.h2, s.t
    return d


def _workspace(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def _as_chw(p: torch.Tensor, Cc: int, N: int) -> torch.Tensor:
    if p.dtype == torch.float32 and p.dim() == 3 and p.shape[0] == Cc and p.shape[1] == N and p.shape[2] == N \
            and p.is_contiguous():
        return p.detach()
    q = p.detach()
    if q.dim() == 2:
        q = q.unsqueeze(0)
    if tuple(q.shape) != (Cc, N, N):
        raise L.PdeError(f"coefficient of shape {tuple(p.shape)} does not match ({Cc},{N},{N})")
    return q.to(torch.float32).contiguous()


#: amplification of rounding error tolerated when a state is rebuilt backwards from a later one
#: (x_{s-1} = (A_s + eps I) x_s grows high-frequency error by up to 1 + 4*coeff per sweep)
CKPT_AMAX = 8.0


def plan_checkpoints(kappa_max: Sequence[float], amax: float = CKPT_AMAX) -> int:
    """Bit s set: keep the state after sweep s as a checkpoint instead of rebuilding it.

    Walking back from the output, the error amplification of the rebuilt state is the product of
    (1 + 4*max coeff) over the sweeps undone since the last exact state; a checkpoint is placed
    whenever that product would pass ``amax``.  Small coefficients (mnist, cifar: 1e-3) need none;
    fashion-like ones (0.27 / 0.54) get one every two sweeps; very large ones one per sweep."""
    mask, amp = 0, 1.0
    for s in range(len(kappa_max) - 1, 0, -1):
        amp *= 1.0 + 4.0 * float(kappa_max[s])
        if amp > amax:
            mask |= 1 << (s - 1)
            amp = 1.0
    return mask


def kappa_max_async(u_like, alpha_base, beta_base, alpha_time_coeff, beta_time_coeff, sweeps: Sequence[Sweep],
                    smooth3=False, clamp_max=None, eps=1e-6):
    """Launch the per-sweep max-coefficient kernel and an asynchronous copy to pinned host memory.
    Returns an object with ``.host`` / ``.event``; the values are valid once ``event.query()`` is True."""
    lib = L.load()
    _require_cuda(u_like, alpha_base, beta_base, alpha_time_coeff, beta_time_coeff)
    B, Cc, N, _ = u_like.shape
    p = [_as_chw(t, Cc, N) for t in (alpha_base, beta_base, alpha_time_coeff, beta_time_coeff)]
    d = _make_desc(max(B, 1), Cc, N, L.PDE_IO_F32, sweeps, smooth3, clamp_max, eps)   # independent of the batch
    kdev = torch.empty(len(sweeps), dtype=torch.float32, device=u_like.device)
    with torch.cuda.device(u_like.device):
        L.check(lib.pde_adi_kappa_max(C.byref(d), *[_ptr(t) for t in p], _ptr(kdev), _stream()), "pde_adi_kappa_max")
        host = torch.empty(len(sweeps), dtype=torch.float32, pin_memory=True)
        host.copy_(kdev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
    return _KmaxOwned(host, ev)


#: entries of the per-device ring of (pinned buffer, event) pairs through which the per-sweep coefficient maxima reach
#: the host; one entry is held from a forward to its backward, so this bounds the PDE-layer calls whose backward is
#: still outstanding (raise it for models with more PDE layers than this in one graph)
KMAX_RING = 256
_kmax_rings = {}


class _KmaxEntry:
    __slots__ = ("host", "event", "gen", "owner")


_kmax_lock = threading.Lock()


def _kmax_channel(n: int):
    """Pinned host buffer and an event for the early copy of the per-sweep coefficient maxima (pde_adi_forward copies
    and records, right behind the factorisation kernel).  Creating a pinned tensor and an event per call costs more
    host time than the launches of a small layer: they come from a ring.  An entry whose previous holder is still alive
    (its backward has not run: more than KMAX_RING PDE-layer calls outstanding — long step groups, activation
    checkpointing, many layers) is left alone and the call gets a buffer and an event of its own instead."""
    if n > L.PDE_MAX_SWEEPS * 4:
        raise L.PdeError(f"{n} coefficient maxima exceed the ring entry of {L.PDE_MAX_SWEEPS * 4}")
    dev = torch.cuda.current_device()
    with _kmax_lock:                                       # autograd threads of several devices, DataParallel replicas
        ring = _kmax_rings.get(dev)
        if ring is None:
            pool = torch.empty(KMAX_RING * L.PDE_MAX_SWEEPS * 4, dtype=torch.float32, pin_memory=True)
            ring = {"next": 0, "entries": []}
            for i in range(KMAX_RING):
                e = _KmaxEntry()
                e.host = pool[i * L.PDE_MAX_SWEEPS * 4:(i + 1) * L.PDE_MAX_SWEEPS * 4]
                e.event = torch.cuda.Event()
                e.event.record()        # creates the underlying hipEvent_t; the library re-records it
                e.gen = 0
                e.owner = None
                ring["entries"].append(e)
            _kmax_rings[dev] = ring
        # the first free entry from `next` on: one long-lived ticket (an output kept without a backward, activation
        # checkpointing) must not make every later call allocate pinned memory and an event of its own
        e = None
        for off in range(KMAX_RING):
            i = (ring["next"] + off) % KMAX_RING
            c = ring["entries"][i]
            if c.owner is None or c.owner() is None:
                e = c
                ring["next"] = (i + 1) % KMAX_RING
                break
        if e is None:                                      # every entry is held by a call whose backward is outstanding
            e = _KmaxEntry()
            e.host = torch.empty(L.PDE_MAX_SWEEPS * 4, dtype=torch.float32, pin_memory=True)
            e.event = torch.cuda.Event()
            e.event.record()
            e.gen = 0
        e.gen += 1
        t = _KmaxTicket(e, n)
        e.owner = weakref.ref(t)
    return t


class _KmaxOwned:
    """Coefficient maxima in a buffer of their own (kappa_max_async)."""
    __slots__ = ("host", "event")

    def __init__(self, host, event):
        self.host, self.event = host, event


def _wait_event(ev, budget=0.02):
    """Wait for an event that is expected within microseconds of the device reaching the factorisation kernel: poll it
    before blocking — a blocking hipEventSynchronize costs the host ~100-150 us to wake up, more than the whole backward
    of a small layer takes to launch.  The poll is bounded by time (in a device-bound loop the host arrives while the
    previous step's backward is still running), not by a number of queries."""
    q = ev.query
    t0 = time.perf_counter()
    while True:
        for _ in range(64):
            if q():
                return
        if time.perf_counter() - t0 > budget:
            break
    ev.synchronize()


class _KmaxTicket:
    """One use of a ring entry: ``host`` / ``event`` as long as the entry has not been handed out again."""
    __slots__ = ("entry", "gen", "n", "__weakref__")

    def __init__(self, entry, n):
        self.entry, self.gen, self.n = entry, entry.gen, n

    def _check(self):
        if self.entry.gen != self.gen:
            raise L.PdeError("the coefficient-maxima ring entry of this call was reused before its backward ran: "
                             "more than functional.KMAX_RING PDE-layer calls are outstanding; raise KMAX_RING")

    @property
    def host(self):
        self._check()
        return self.entry.host[:self.n]

    @property
    def event(self):
        self._check()
        return self.entry.event

    def wait(self):
        """Values once the copy has landed (waits for the factorisation kernel only)."""
        _wait_event(self.event)
        return self.host.tolist()


class _KmaxConcat:
    """The maxima of several calls in call order (a layer composed from per-step calls): ``host`` / ``event`` like a ticket."""
    __slots__ = ("parts",)

    def __init__(self, parts):
        self.parts = list(parts)

    class _Ev:
        def __init__(self, evs):
            self.evs = evs

        def query(self):
            return all(e.query() for e in self.evs)

        def synchronize(self):
            for e in self.evs:
                e.synchronize()

    @property
    def event(self):
        return _KmaxConcat._Ev([p.event for p in self.parts])

    @property
    def host(self):
        return torch.cat([p.host for p in self.parts])

    def wait(self):
        return [v for p in self.parts for v in p.wait()]


class _AdiFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, ab, bb, asl, bsl, sweeps, smooth3, clamp_max, eps, ckpt, kmax_sink):
        lib = L.load()
        _require_cuda(u, ab, bb, asl, bsl)
        if u.dim() != 4 or u.shape[2] != u.shape[3]:
            raise L.PdeError(f"expected (B,C,N,N), got {tuple(u.shape)}")
        B, Cc, N, _ = u.shape
        if u.dtype not in (torch.float32, torch.bfloat16):
            u = u.float()
        u = u.contiguous()
        p = [_as_chw(t, Cc, N) for t in (ab, bb, asl, bsl)]
        d = _make_desc(B, Cc, N, _io_dtype(u), sweeps, smooth3, clamp_max, eps)
        y = torch.empty_like(u)
        nbytes = lib.pde_adi_forward_workspace_bytes(C.byref(d))
        ws = _workspace(nbytes, u.device)
        need_grad = any(ctx.needs_input_grad[:5])
        want_kmax = need_grad and (ckpt == "auto" or kmax_sink is not None)
        kdev = torch.empty(len(sweeps), dtype=torch.float32, device=u.device) if want_kmax else None
        with torch.cuda.device(u.device):
            # per-sweep maximum coefficient (a by-product of the factorisation kernel): copied to pinned host
            # memory right behind that kernel, before the sweep launch, so whoever plans checkpoints from it
            # waits for the factorisation only
            tk = _kmax_channel(len(sweeps)) if want_kmax else None
            L.check(lib.pde_adi_forward(C.byref(d), _ptr(u), _ptr(y), *[_ptr(t) for t in p], _ptr(kdev),
                                        _ptr(tk.host if tk else None), C.c_void_p(tk.event.cuda_event if tk else 0),
                                        _ptr(ws), ws.numel(), _stream()), "pde_adi_forward")
            ctx.kmax = tk
            if want_kmax and kmax_sink is not None:
                kmax_sink.append(tk)
        ctx.fwd_ws = ws if need_grad else None       # factorisation reused by the backward
        ctx.save_for_backward(y, u if (need_grad and ckpt != 0) else None, *p)
        ctx.cfg = (sweeps, smooth3, clamp_max, eps, ckpt)
        ctx.param_shapes = [t.shape for t in (ab, bb, asl, bsl)]
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = L.load()
        y, u, *p = ctx.saved_tensors
        sweeps, smooth3, clamp_max, eps, ckpt = ctx.cfg
        B, Cc, N, _ = y.shape
        gy = gy.to(y.dtype).contiguous()
        d = _make_desc(B, Cc, N, _io_dtype(y), sweeps, smooth3, clamp_max, eps)
        gu = torch.empty_like(y)
        gp = [torch.empty_like(t) for t in p]
        if ckpt == "auto":
            bits = plan_checkpoints(ctx.kmax.wait())
        else:
            bits = int(ckpt)
        mask = (C.c_uint64 * 2)(bits & (2 ** 64 - 1), bits >> 64)
        nbytes = lib.pde_adi_backward_workspace_bytes(C.byref(d), bin(bits).count("1"))
        ws = _workspace(nbytes, y.device)
        with torch.cuda.device(y.device):          # autograd thread: set device, fetch the stream here
            L.check(lib.pde_adi_backward(C.byref(d), _ptr(gy), _ptr(y), _ptr(u if bits else None), mask, _ptr(gu),
                                         *[_ptr(t) for t in p], *[_ptr(t) for t in gp], _ptr(ctx.fwd_ws),
                                         _ptr(ws), ws.numel(), _stream()), "pde_adi_backward")
        ctx.fwd_ws = None
        gp = [g.reshape(s) for g, s in zip(gp, ctx.param_shapes)]
        return (gu, *gp, None, None, None, None, None, None)


class _AdiMixedFn(torch.autograd.Function):
    """One call of a layer whose time steps are separated by a channel operator: cifar10.py:84-112 (``mode``
    "pre": u <- M u, then the step's sweeps) and SVHN.py:55-72 ("post": the sweeps, then u <- K u).

    One factorisation for the whole schedule, then per step one mixing launch and one sweep launch; the
    backward walks the steps in reverse, accumulating the partial sums of the parameter gradients and of the
    matrix gradient on the device, and reduces them once at the end (instead of a torch autograd node, a
    factorisation, a reduction and a gradient-accumulation add per step)."""

    @staticmethod
    def forward(ctx, u, ab, bb, asl, bsl, M, steps, mode, smooth3, clamp_max, eps, ckpt, kmax_sink):
        lib = L.load()
        _require_cuda(u, ab, bb, asl, bsl, M)
        if u.dim() != 4 or u.shape[2] != u.shape[3]:
            raise L.PdeError(f"expected (B,C,N,N), got {tuple(u.shape)}")
        B, Cc, N, _ = u.shape
        if u.dtype not in (torch.float32, torch.bfloat16):
            u = u.float()
        u = u.contiguous()
        sps, K = len(steps[0]), len(steps)
        sweeps = tuple(s for st in steps for s in st)
        p = [_as_chw(t, Cc, N) for t in (ab, bb, asl, bsl)]
        Mf = M.detach().to(torch.float32).contiguous()
        d = _make_desc(B, Cc, N, _io_dtype(u), sweeps, smooth3, clamp_max, eps)
        sws = _workspace(lib.pde_adi_steps_workspace_bytes(C.byref(d), sps), u.device)
        need_grad = any(ctx.needs_input_grad[:6])
        want_kmax = need_grad and (ckpt == "auto" or kmax_sink is not None)
        kdev = torch.empty(len(sweeps), dtype=torch.float32, device=u.device) if want_kmax else None
        # states[2k]: output of step k's first operator, states[2k+1]: of its second (= input of step k+1); the last of them
        # is the layer output and lives in a tensor of its own (no copy, and nobody can reach the kept states through it)
        states = torch.empty((2 * K - 1,) + tuple(u.shape), dtype=u.dtype, device=u.device)
        y = torch.empty_like(u)
        with torch.cuda.device(u.device):
            tk = _kmax_channel(len(sweeps)) if want_kmax else None
            L.check(lib.pde_adi_mixed_forward(C.byref(d), sps, 1 if mode == "pre" else 2, _ptr(u), _ptr(states), _ptr(y), _ptr(Mf),
                                              *[_ptr(t) for t in p], _ptr(kdev), _ptr(tk.host if tk else None),
                                              C.c_void_p(tk.event.cuda_event if tk else 0),
                                              _ptr(sws), sws.numel(), _stream()),
                    "pde_adi_mixed_forward")
            ctx.kmax = tk
            if want_kmax and kmax_sink is not None:
                kmax_sink.append(tk)
        if need_grad:
            ctx.save_for_backward(u, states, y, Mf, *p)
            ctx.sws = sws
        ctx.cfg = (steps, mode, smooth3, clamp_max, eps, ckpt)
        ctx.param_shapes = [t.shape for t in (ab, bb, asl, bsl)]
        ctx.M_dtype = M.dtype
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = L.load()
        u, states, y, Mf, *p = ctx.saved_tensors
        steps, mode, smooth3, clamp_max, eps, ckpt = ctx.cfg
        B, Cc, N, _ = u.shape
        sps, K, HW = len(steps[0]), len(steps), N * N
        sweeps = tuple(s for st in steps for s in st)
        d = _make_desc(B, Cc, N, _io_dtype(u), sweeps, smooth3, clamp_max, eps)
        if ckpt == "auto":
            km = ctx.kmax.wait()
            bits = 0
            for k in range(K):                           # one step-local mask for every step: the union
                bits |= plan_checkpoints(km[k * sps:(k + 1) * sps])
        else:
            bits = int(ckpt)
        nck = bin(bits).count("1")
        mask = (C.c_uint64 * 2)(bits & (2 ** 64 - 1), bits >> 64)
        ws = _workspace(lib.pde_adi_mixed_backward_workspace_bytes(C.byref(d), sps, nck), u.device)
        g_in = gy.to(u.dtype).contiguous()
        g_a = torch.empty_like(g_in)
        gp = [torch.empty_like(t) for t in p]
        gM = torch.empty_like(Mf)
        with torch.cuda.device(u.device):
            L.check(lib.pde_adi_mixed_backward(C.byref(d), sps, 1 if mode == "pre" else 2, _ptr(g_in), _ptr(u), _ptr(states),
                                               _ptr(y), _ptr(Mf), mask, _ptr(g_a), *[_ptr(t) for t in p], *[_ptr(t) for t in gp],
                                               _ptr(gM), _ptr(ctx.sws), _ptr(ws), ws.numel(), _stream()),
                    "pde_adi_mixed_backward")
        gp = [g.reshape(s) for g, s in zip(gp, ctx.param_shapes)]
        return (g_a, *gp, gM.to(ctx.M_dtype), None, None, None, None, None, None, None)


class _AdiSmallFn(torch.autograd.Function):
    """A layer with a channel operator between its time steps at C <= 4, whole time loop in ONE launch per pass
    (pde_adi_small_*): cifar10.py:84-112 ("pre"), SVHN.py:55-76 ("post", with the skip blend when ``skip_weight``
    is given).  The sweep output of every step is kept for the backward, as autograd keeps it in the reference."""

    @staticmethod
    def forward(ctx, u, ab, bb, asl, bsl, M, skip_weight, steps, mode, smooth3, clamp_max, eps, ckpt, kmax_sink):
        lib = L.load()
        _require_cuda(u, ab, bb, asl, bsl, M, skip_weight)
        B, Cc, N, _ = u.shape
        if u.dtype not in (torch.float32, torch.bfloat16):
            u = u.float()
        u = u.contiguous()
        sps, K = len(steps[0]), len(steps)
        sweeps = steps.flat
        p = [_as_chw(t, Cc, N) for t in (ab, bb, asl, bsl)]
        Mf = M.detach().to(torch.float32).contiguous()
        sw = None if skip_weight is None else skip_weight.detach().to(torch.float32).reshape(1).contiguous()
        d = _make_desc(B, Cc, N, _io_dtype(u), sweeps, smooth3, clamp_max, eps)
        sws = _workspace(lib.pde_adi_steps_workspace_bytes(C.byref(d), sps), u.device)
        need_grad = any(ctx.needs_input_grad[:7])
        want_kmax = need_grad and (ckpt == "auto" or kmax_sink is not None)
        kdev = torch.empty(len(sweeps), dtype=torch.float32, device=u.device) if want_kmax else None
        states = torch.empty((K,) + tuple(u.shape), dtype=u.dtype, device=u.device) if need_grad else None
        y = torch.empty_like(u)
        with torch.cuda.device(u.device):
            tk = _kmax_channel(len(sweeps)) if want_kmax else None
            L.check(lib.pde_adi_small_forward(C.byref(d), sps, 1 if mode == "pre" else 2, _ptr(u), _ptr(y), _ptr(states),
                                              _ptr(Mf), _ptr(sw), *[_ptr(t) for t in p], _ptr(kdev),
                                              _ptr(tk.host if tk else None), C.c_void_p(tk.event.cuda_event if tk else 0),
                                              _ptr(sws), sws.numel(), _stream()), "pde_adi_small_forward")
            ctx.kmax = tk
            if want_kmax and kmax_sink is not None:
                kmax_sink.append(tk)
        if need_grad:
            ctx.save_for_backward(u, states, Mf, sw, *p)
            ctx.sws = sws
        ctx.cfg = (steps, mode, smooth3, clamp_max, eps, ckpt)
        ctx.param_shapes = [t.shape for t in (ab, bb, asl, bsl)]
        ctx.M_dtype = M.dtype
        ctx.skip_meta = None if skip_weight is None else (skip_weight.dtype, skip_weight.shape)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = L.load()
        u, states, Mf, sw, *p = ctx.saved_tensors
        steps, mode, smooth3, clamp_max, eps, ckpt = ctx.cfg
        B, Cc, N, _ = u.shape
        sps, K = len(steps[0]), len(steps)
        sweeps = steps.flat
        d = _make_desc(B, Cc, N, _io_dtype(u), sweeps, smooth3, clamp_max, eps)
        if ckpt == "auto":
            km = ctx.kmax.wait()
            bits = 0
            for k in range(K):                           # one step-local mask for every step: the union
                bits |= plan_checkpoints(km[k * sps:(k + 1) * sps])
        else:
            bits = int(ckpt)
        mask = (C.c_uint64 * 2)(bits & (2 ** 64 - 1), bits >> 64)
        ws = _workspace(lib.pde_adi_small_backward_workspace_bytes(C.byref(d), sps, bin(bits).count("1")), u.device)
        g_in = gy.to(u.dtype).contiguous()
        gu = torch.empty_like(g_in)
        gp = [torch.empty_like(t) for t in p]
        gM = torch.empty_like(Mf)
        gsw = torch.empty(1, dtype=torch.float32, device=u.device) if sw is not None else None
        with torch.cuda.device(u.device):
            L.check(lib.pde_adi_small_backward(C.byref(d), sps, 1 if mode == "pre" else 2, _ptr(g_in), _ptr(u), _ptr(states),
                                               _ptr(Mf), _ptr(sw), mask, _ptr(gu), *[_ptr(t) for t in p],
                                               *[_ptr(t) for t in gp], _ptr(gM), _ptr(gsw), _ptr(ctx.sws), _ptr(ws),
                                               ws.numel(), _stream()), "pde_adi_small_backward")
        gp = [g.reshape(s) for g, s in zip(gp, ctx.param_shapes)]
        gskip = None if sw is None else gsw.to(ctx.skip_meta[0]).reshape(ctx.skip_meta[1])
        return (gu, *gp, gM.to(ctx.M_dtype), gskip, None, None, None, None, None, None, None)


class _AdiMultiFn(torch.autograd.Function):
    """Several mixing-first layers (cifar10.EnhancedDiffusionLayer / cifar_2version.LearnableDiffusionLayer, C <= 4)
    on the SAME input in one launch per pass (pde_adi_multi_*): cifar10.py:272-280, cifar_2version.py:287-288.
    Returns (sum_i w_i y_i, y_1, ..., y_L); the y_i are the last sweep outputs the kernel keeps anyway."""

    @staticmethod
    def forward(ctx, u, weights, specs, want_sums, ckpt, *flat):
        lib = L.load()
        nl = len(specs)
        _require_cuda(u, weights, *flat)
        B, Cc, N, _ = u.shape
        if u.dtype not in (torch.float32, torch.bfloat16):
            u = u.float()
        u = u.contiguous()
        need_grad = any(ctx.needs_input_grad)
        want_kmax = need_grad and ckpt == "auto"
        if ckpt != "auto" and not isinstance(ckpt, (tuple, list)):
            ckpt = (int(ckpt),) * nl                      # one step-local mask for every layer
        if ckpt != "auto" and len(ckpt) != nl:
            raise L.PdeError(f"adi_diffuse_multi: {len(ckpt)} checkpoint masks for {nl} layers")
        ctx.ckpt = ckpt
        ctx.set_materialize_grads(False)
        wdev = None if weights is None else weights.detach().to(torch.float32).contiguous()   # read by the kernels on the device
        arr = (L.PdeSmallLayer * nl)()
        keep, descs, per, sums = [], [], [], []
        for i, (steps, smooth3, clamp_max, eps) in enumerate(specs):
            ab, bb, asl, bsl, M = flat[5 * i:5 * i + 5]
            sps, K = len(steps[0]), len(steps)
            sweeps = steps.flat
            p = [_as_chw(t, Cc, N) for t in (ab, bb, asl, bsl)]
            Mf = M.detach().to(torch.float32).contiguous()
            d = _make_desc(B, Cc, N, _io_dtype(u), sweeps, smooth3, clamp_max, eps)
            sws = _workspace(lib.pde_adi_steps_workspace_bytes(C.byref(d), sps), u.device)
            states = torch.empty((K,) + tuple(u.shape), dtype=u.dtype, device=u.device)
            kdev = torch.empty(len(sweeps), dtype=torch.float32, device=u.device) if want_kmax else None
            with torch.cuda.device(u.device):
                tk = _kmax_channel(len(sweeps)) if want_kmax else None
            host = tk.host if tk else None
            a = arr[i]
            a.desc, a.sweeps_per_step, a.mode = C.pointer(d), sps, 1
            a.M = Mf.data_ptr()
            a.alpha_base, a.beta_base, a.alpha_slope, a.beta_slope = (t.data_ptr() for t in p)
            a.weight = 0.0
            a.weight_ptr = None if wdev is None else wdev.data_ptr() + 4 * i
            a.states = states.data_ptr()
            a.steps_workspace, a.steps_workspace_bytes = sws.data_ptr(), sws.numel()
            a.kappa_max = kdev.data_ptr() if kdev is not None else None
            a.kappa_max_host = host.data_ptr() if host is not None else None
            psum = torch.empty((B, Cc), dtype=torch.float32, device=u.device) if want_sums else None
            a.plane_sums = psum.data_ptr() if psum is not None else None
            sums.append(psum)
            keep.append((p, Mf, sws, states, kdev, tk))
            descs.append(d)
            per.append((sps, K, [t.shape for t in (ab, bb, asl, bsl)], M.dtype))
        out = torch.empty_like(u)
        with torch.cuda.device(u.device):
            ev = keep[-1][5].event if want_kmax else None      # recorded behind the last layer's copy
            L.check(lib.pde_adi_multi_forward(nl, arr, _ptr(u), _ptr(out), C.c_void_p(ev.cuda_event if ev is not None else 0),
                                              _stream()), "pde_adi_multi_forward")
        # the input, the parameter views and the operators go through save_for_backward (an in-place change of any of them
        # between forward and backward then raises torch's version-counter error instead of giving gradients that mix the
        # forward's factorisation with new parameter values); workspaces, states and tickets stay on ctx
        saved = [u] + ([wdev] if wdev is not None else [])
        for k in keep:
            saved += list(k[0]) + [k[1]]
        ctx.save_for_backward(*saved)
        ctx.keep, ctx.descs, ctx.per = [k[2:] for k in keep], descs, per
        ctx.has_w = weights is not None
        ctx.want_sums = want_sums
        ys = [k[3][-1] for k in keep]                      # the last sweep output of every layer
        return (out, *ys, *sums) if want_sums else (out, *ys)

    @staticmethod
    def backward(ctx, gout, *gall):
        lib = L.load()
        nl = len(ctx.per)
        gys, gsums = gall[:nl], (gall[nl:] if ctx.want_sums else (None,) * nl)
        saved = ctx.saved_tensors
        u = saved[0]
        wdev = saved[1] if ctx.has_w else None
        pm = saved[2 if ctx.has_w else 1:]                 # per layer: four parameter views and the operator
        B, Cc, N, _ = u.shape
        arr = (L.PdeSmallLayer * nl)()
        if ctx.ckpt == "auto":
            _wait_event(ctx.keep[-1][3].event)
        outs, hold = [], []
        gout_c = None if gout is None else gout.to(u.dtype).contiguous()
        for i in range(nl):
            sws, states, kdev, tk = ctx.keep[i]
            p, Mf = pm[5 * i:5 * i + 4], pm[5 * i + 4]
            sps, K, shapes, Mdt = ctx.per[i]
            d = ctx.descs[i]
            bits = 0
            if ctx.ckpt == "auto":
                km = tk.host.tolist()
                for k in range(K):
                    bits |= plan_checkpoints(km[k * sps:(k + 1) * sps])
            else:
                bits = int(ctx.ckpt[i])
            mask = (C.c_uint64 * 2)(bits & (2 ** 64 - 1), bits >> 64)
            ws = _workspace(lib.pde_adi_small_backward_workspace_bytes(C.byref(d), sps, bin(bits).count("1")), u.device)
            gp = [torch.empty_like(t) for t in p]
            gM = torch.empty_like(Mf)
            gw = torch.empty(1, dtype=torch.float32, device=u.device)
            gyi = None if gys[i] is None else gys[i].to(u.dtype).contiguous()
            gsi = None if gsums[i] is None else gsums[i].to(torch.float32).contiguous()
            a = arr[i]
            a.desc, a.sweeps_per_step, a.mode = C.pointer(d), sps, 1
            a.M = Mf.data_ptr()
            a.alpha_base, a.beta_base, a.alpha_slope, a.beta_slope = (t.data_ptr() for t in p)
            a.weight = 0.0
            a.weight_ptr = None if wdev is None else wdev.data_ptr() + 4 * i
            a.states = states.data_ptr()
            a.steps_workspace, a.steps_workspace_bytes = sws.data_ptr(), sws.numel()
            a.gys = gyi.data_ptr() if gyi is not None else None
            a.g_plane_sums = gsi.data_ptr() if gsi is not None else None
            a.ckpt_mask = mask
            a.g_alpha_base, a.g_beta_base, a.g_alpha_slope, a.g_beta_slope = (t.data_ptr() for t in gp)
            a.gM, a.g_weight = gM.data_ptr(), gw.data_ptr()
            a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
            hold.append((mask, ws, gyi, gsi))
            outs.append((gp, gM, gw, shapes, Mdt))
        if gout_c is None and all(h[2] is None and h[3] is None for h in hold):
            raise L.PdeError("adi_diffuse_multi: no incoming gradient")
        gu = torch.empty_like(u)
        with torch.cuda.device(u.device):
            L.check(lib.pde_adi_multi_backward(nl, arr, _ptr(gout_c), _ptr(u), _ptr(gu), _stream()), "pde_adi_multi_backward")
        flat = []
        for gp, gM, gw, shapes, Mdt in outs:
            flat += [g.reshape(s) for g, s in zip(gp, shapes)] + [gM.to(Mdt)]
        gweights = torch.cat([o[2] for o in outs]) if ctx.has_w else None
        return (gu, gweights, None, None, None, *flat)


def adi_diffuse_multi(u, layers, weights=None, plane_sums=False, checkpoints="auto"):
    """Run several mixing-first layers on the same ``u`` in one launch per pass.

    ``layers``: list of dicts with keys alpha_base, beta_base, alpha_time_coeff, beta_time_coeff, M, steps and
    optionally smooth3, clamp_max, eps.  ``weights`` (L,): ``out = sum_i weights[i] * y_i`` (cifar10.py:277-280
    without the attention gates); None: ``out`` is not meaningful.  Returns ``(out, [y_1 .. y_L])``, and with
    ``plane_sums`` also ``[s_1 .. s_L]``, ``s_i[b,c] = sum_hw y_i`` — the adaptive average pool of
    cifar10.py:239 times H*W, a by-product of the kernel (differentiable like the ``y_i``).  ``checkpoints``: "auto",
    one step-local bit mask for every layer (an int) or one mask per layer (a tuple) — with masks there is no host wait
    at all and the call is hipGraph-capturable."""
    specs = tuple((_as_schedule(ly["steps"]), bool(ly.get("smooth3", False)), ly.get("clamp_max"),
                   float(ly.get("eps", 1e-6))) for ly in layers)
    flat = []
    for ly in layers:
        flat += [ly["alpha_base"], ly["beta_base"], ly["alpha_time_coeff"], ly["beta_time_coeff"], ly["M"]]
    if isinstance(checkpoints, list):
        checkpoints = tuple(checkpoints)
    nl = len(layers)
    H = L.host_ext()
    if H is not None and u.dim() == 4 and u.is_cuda and u.shape[0] > 0 and len({len(sp[0][0]) for sp in specs}) == 1:
        # the native host path (csrc/host_ext.cpp): the same call of the C ABI from a C++ autograd node
        if checkpoints == "auto":
            mode, masks = 1, []
        else:
            masks = [int(checkpoints)] * nl if not isinstance(checkpoints, tuple) else [int(c) for c in checkpoints]
            if len(masks) != nl:
                raise L.PdeError(f"adi_diffuse_multi: {len(masks)} checkpoint masks for {nl} layers")
            mode = 0
        B, Cc, N, _ = u.shape
        io = L.PDE_IO_BF16 if u.dtype == torch.bfloat16 else L.PDE_IO_F32
        addrs = [C.addressof(_make_desc(B, Cc, N, io, st.flat, sm, cm, ep)) for st, sm, cm, ep in specs]
        res = H.multi(u, weights, flat, addrs, len(specs[0][0][0]), bool(plane_sums), mode, masks, CKPT_AMAX)
        if plane_sums:
            return res[0], list(res[1:1 + nl]), list(res[1 + nl:])
        return res[0], list(res[1:])
    res = _AdiMultiFn.apply(u, weights, specs, bool(plane_sums), checkpoints, *flat)
    if plane_sums:
        return res[0], list(res[1:1 + nl]), list(res[1 + nl:])
    return res[0], list(res[1:])


_small_ok_cache = {}


def adi_small_supported(u, steps, smooth3=False, clamp_max=None, eps=1e-6) -> bool:
    """True when ``adi_diffuse_small`` can run this layer call (C <= 4, N in {16, 28, 32}, Strang or Lie steps)."""
    if u.dim() != 4 or u.shape[2] != u.shape[3] or not u.is_cuda:
        return False
    B, Cc, N, _ = u.shape
    if B == 0 or Cc > 4 or N > L.PDE_MAX_N or len(steps) * len(steps[0]) > L.PDE_MAX_SWEEPS:
        return False
    io = L.PDE_IO_BF16 if u.dtype == torch.bfloat16 else L.PDE_IO_F32
    steps = _as_schedule(steps)
    key = (B, Cc, N, io, steps.flat, bool(smooth3), clamp_max, float(eps), len(steps[0]))
    ok = _small_ok_cache.get(key)
    if ok is None:
        if len(_small_ok_cache) > 256:
            _small_ok_cache.clear()
        d = _make_desc(B, Cc, N, io, steps.flat, smooth3, clamp_max, eps)
        ok = _small_ok_cache[key] = bool(L.load().pde_adi_small_supported(C.byref(d), len(steps[0])))
    return ok


def adi_diffuse_small(u, alpha_base, beta_base, alpha_time_coeff, beta_time_coeff, M, steps, mode: str, skip_weight=None,
                      smooth3: bool = False, clamp_max: Optional[float] = None, eps: float = 1e-6, checkpoints="auto",
                      kmax_sink: Optional[list] = None):
    """``adi_diffuse_mixed`` (plus the SVHN skip blend when ``skip_weight`` is given) for C <= 4 channels: the whole
    time loop in one launch forward and one backward.  Check ``adi_small_supported`` first."""
    if mode not in ("pre", "post"):
        raise ValueError(mode)
    steps = _as_schedule(steps)
    if kmax_sink is None and (checkpoints == "auto" or isinstance(checkpoints, int)) and u.shape[0] > 0:
        H = L.host_ext()
        if H is not None and u.dim() == 4 and u.is_cuda:
            # the native host path (csrc/host_ext.cpp): the same call of the C ABI from a C++ autograd node
            B, Cc, N, _ = u.shape
            d = _make_desc(B, Cc, N, L.PDE_IO_BF16 if u.dtype == torch.bfloat16 else L.PDE_IO_F32, steps.flat, smooth3,
                           clamp_max, eps)
            return H.small(u, alpha_base, beta_base, alpha_time_coeff, beta_time_coeff, M, skip_weight, C.addressof(d),
                           len(steps[0]), 1 if mode == "pre" else 2, 1 if checkpoints == "auto" else 0,
                           0 if checkpoints == "auto" else int(checkpoints), CKPT_AMAX)
    return _AdiSmallFn.apply(u, alpha_base, beta_base, alpha_time_coeff, beta_time_coeff, M, skip_weight, steps, mode,
                             bool(smooth3), clamp_max, float(eps), checkpoints, kmax_sink)


def adi_diffuse_mixed(u, alpha_base, beta_base, alpha_time_coeff, beta_time_coeff, M, steps, mode: str,
                      smooth3: bool = False, clamp_max: Optional[float] = None, eps: float = 1e-6, checkpoints="auto",
                      kmax_sink: Optional[list] = None):
    """All time steps of a layer with a channel operator ``M`` between them, as one autograd node.

    ``steps``: list of per-step sweep lists (``adi_schedule``); ``mode`` "pre": ``u <- M u`` before every
    step (cifar10.py:91), "post": after every step (SVHN.py:71).  ``checkpoints``: "auto" or a bit mask
    relative to a step (bit 0 = state after the step's first sweep), applied to every step."""
    if mode not in ("pre", "post"):
        raise ValueError(mode)
    if u.shape[0] == 0:
        return _empty_passthrough(u, alpha_base, beta_base, alpha_time_coeff, beta_time_coeff, M)
    steps = tuple(tuple(st) for st in steps)
    if len({len(st) for st in steps}) != 1:
        raise ValueError("every step must have the same number of sweeps")
    if u.dim() == 4 and u.is_cuda and L.load().pde_adi_line_length_path(int(u.shape[-1])) == 2:
        # a line length without fused kernels (pde_adi_line_length_path: any size up to PDE_MAX_N_GENERIC): the per-step
        # entry points do not exist there; compose the layer from its own pieces — the channel operator and the sweeps of
        # one step per call, chained by autograd (the step-local checkpoint mask applies to every step unchanged)
        tickets = [] if kmax_sink is not None else None
        for st in steps:
            if mode == "pre":
                u = channel_mix(u, M)
            u = adi_diffuse(u, alpha_base, beta_base, alpha_time_coeff, beta_time_coeff, st, smooth3=smooth3,
                            clamp_max=clamp_max, eps=eps, checkpoints=checkpoints, kmax_sink=tickets)
            if mode == "post":
                u = channel_mix(u, M)
        if tickets and len(tickets) == len(steps):           # the whole layer's maxima, step after step (lagged plans)
            kmax_sink.append(_KmaxConcat(tickets))
        return u
    if kmax_sink is None and (checkpoints == "auto" or isinstance(checkpoints, int)):
        H = L.host_ext()
        if H is not None and u.dim() == 4 and u.is_cuda and u.shape[2] == u.shape[3]:
            # the native host path (csrc/host_ext.cpp): the same two calls of the C ABI from a C++ autograd node
            B, Cc, N, _ = u.shape
            flat = tuple(s for st in steps for s in st)
            d = _make_desc(B, Cc, N, L.PDE_IO_BF16 if u.dtype == torch.bfloat16 else L.PDE_IO_F32, flat, smooth3, clamp_max, eps)
            return H.mixed(u, alpha_base, beta_base, alpha_time_coeff, beta_time_coeff, M, C.addressof(d), len(steps[0]),
                           1 if mode == "pre" else 2, 1 if checkpoints == "auto" else 0,
                           0 if checkpoints == "auto" else int(checkpoints), CKPT_AMAX)
    return _AdiMixedFn.apply(u, alpha_base, beta_base, alpha_time_coeff, beta_time_coeff, M, steps, mode, bool(smooth3),
                             clamp_max, float(eps), checkpoints, kmax_sink)


def _empty_passthrough(u, *params):
    """Empty batch: nothing to launch.  Like the reference's torch ops, pass the empty tensor through and
    keep it connected to the parameters (their gradients are zeros, not None)."""
    _require_cuda(u, *params)
    tie = sum((p.sum() for p in params if isinstance(p, torch.Tensor)), u.new_zeros(()))
    return u + 0 * tie.to(u.dtype)


def adi_diffuse(u, alpha_base, beta_base, alpha_time_coeff, beta_time_coeff, sweeps: Sequence[Sweep],
                smooth3: bool = False, clamp_max: Optional[float] = None, eps: float = 1e-6, checkpoints="auto",
                kmax_sink: Optional[list] = None):
    """Run ``sweeps`` (a flat list) of implicit diffusion on ``u`` (B,C,N,N) in one fused launch.

    Replaces the reference's time loop over diffuse_x/diffuse_y/thomas_solver_batch
    (mnist_test.py:44-198, cifar10.py:74-211) and its autograd backward.

    ``checkpoints``: "auto" (default) chooses the backward's checkpoints from the coefficients
    (``plan_checkpoints``); an int is an explicit bit mask (0: rebuild every state from the output).
    ``kmax_sink``: a list that receives an object with ``.host`` (pinned tensor) and ``.event``: the per-sweep
    maximum coefficient of this call (valid once the event has completed).
    """
    if u.shape[0] == 0:
        return _empty_passthrough(u, alpha_base, beta_base, alpha_time_coeff, beta_time_coeff)
    if not isinstance(sweeps, tuple):
        sweeps = tuple(sweeps)
    if (kmax_sink is None and (checkpoints == "auto" or isinstance(checkpoints, int))) or \
            (kmax_sink is not None and isinstance(checkpoints, int)):
        H = L.host_ext()
        if H is not None and u.dim() == 4 and u.is_cuda and u.dtype in (torch.float32, torch.bfloat16, torch.float16,
                                                                        torch.float64):
            # the native host path (csrc/host_ext.cpp): the same two calls of the C ABI from a C++ autograd node
            B, Cc, N, _ = u.shape
            d = _make_desc(B, Cc, N, L.PDE_IO_BF16 if u.dtype == torch.bfloat16 else L.PDE_IO_F32, sweeps, smooth3,
                           clamp_max, eps)
            if checkpoints == "auto":
                mode, lo, hi = 1, 0, 0
            else:
                bits = int(checkpoints)
                mode, lo, hi = 0, bits & _M64, (bits >> 64) & _M64
                lo, hi = (lo - (1 << 64) if lo >> 63 else lo), (hi - (1 << 64) if hi >> 63 else hi)
            if kmax_sink is not None:                         # "lagged": an explicit mask now, the maxima for the next plan
                y, tk = H.adi_lagged(u, alpha_base, beta_base, alpha_time_coeff, beta_time_coeff, C.addressof(d), lo, hi)
                if tk is not None:
                    kmax_sink.append(tk)
                return y
            return H.adi(u, alpha_base, beta_base, alpha_time_coeff, beta_time_coeff, C.addressof(d), mode, lo, hi, CKPT_AMAX)
    return _AdiFn.apply(u, alpha_base, beta_base, alpha_time_coeff, beta_time_coeff,
                        sweeps, bool(smooth3), clamp_max, float(eps), checkpoints, kmax_sink)


# --------------------------------------------------------------------------- channel mixing
class _MixFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, M):
        lib = L.load()
        _require_cuda(u, M)
        B, Cc = u.shape[0], u.shape[1]
        HW = u[0, 0].numel()
        if u.dtype not in (torch.float32, torch.bfloat16):
            u = u.float()
        u = u.contiguous()
        Mf = M.detach().to(torch.float32).contiguous()
        out = torch.empty_like(u)
        with torch.cuda.device(u.device):
            L.check(lib.pde_channel_mix_forward(B, Cc, HW, _io_dtype(u), _ptr(u), _ptr(Mf), _ptr(out), _stream()),
                    "pde_channel_mix_forward")
        ctx.save_for_backward(u, Mf)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = L.load()
        u, Mf = ctx.saved_tensors
        B, Cc = u.shape[0], u.shape[1]
        HW = u[0, 0].numel()
        gout = gout.to(u.dtype).contiguous()
        gu = torch.empty_like(u)
        gM = torch.empty_like(Mf)
        ws = _workspace(lib.pde_channel_mix_backward_workspace_bytes(B, Cc, HW), u.device)
        with torch.cuda.device(u.device):
            L.check(lib.pde_channel_mix_backward(B, Cc, HW, _io_dtype(u), _ptr(u), _ptr(gout), _ptr(Mf), _ptr(gu),
                                                 _ptr(gM), _ptr(ws), ws.numel(), _stream()),
                    "pde_channel_mix_backward")
        return gu, gM


def channel_mix(u, M):
    """out[b,i,p] = sum_j M[i,j] u[b,j,p] — cifar10.py:65-72 and SVHN.py:78-86."""
    if u.shape[0] == 0:
        return _empty_passthrough(u, M)
    return _MixFn.apply(u, M)


# --------------------------------------------------------------------------- BatchNorm2d + 4x4 avg/max pooling
class _BnPoolFn(torch.autograd.Function):
    """cifar10.py:346-353 in two passes over the activation (pde_tail.hip): statistics, then normalise + pool."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps):
        lib = L.load()
        _require_cuda(x, gamma, beta)
        B, Cc, N, _ = x.shape
        xf = x.contiguous()
        gm = None if gamma is None else gamma.detach().to(torch.float32).contiguous()
        bt = None if beta is None else beta.detach().to(torch.float32).contiguous()
        mean = torch.empty(Cc, dtype=torch.float32, device=x.device)
        invstd = torch.empty_like(mean)
        out = torch.empty((B, 2 * Cc, 4, 4), dtype=torch.float32, device=x.device)
        amax = torch.empty((B, Cc, 4, 4), dtype=torch.int32, device=x.device)
        ws = _workspace(lib.pde_bn_pool_workspace_bytes(B, Cc), x.device)
        with torch.cuda.device(x.device):
            L.check(lib.pde_bn_pool_forward(B, Cc, N, _ptr(xf), _ptr(gm), _ptr(bt), float(eps), 1 if training else 0,
                                            float(momentum), _ptr(running_mean), _ptr(running_var), _ptr(mean), _ptr(invstd),
                                            _ptr(out), _ptr(amax), _ptr(ws), ws.numel(), _stream()), "pde_bn_pool_forward")
        ctx.save_for_backward(xf, gm, mean, invstd, amax)
        ctx.training = bool(training)
        ctx.has = (gamma is not None, beta is not None)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = L.load()
        xf, gm, mean, invstd, amax = ctx.saved_tensors
        B, Cc, N, _ = xf.shape
        g = gout.to(torch.float32).contiguous()
        gx = torch.empty_like(xf)
        gg = torch.empty(Cc, dtype=torch.float32, device=xf.device)
        gb = torch.empty_like(gg)
        ws = _workspace(lib.pde_bn_pool_workspace_bytes(B, Cc), xf.device)
        with torch.cuda.device(xf.device):
            L.check(lib.pde_bn_pool_backward(B, Cc, N, _ptr(xf), _ptr(gm), _ptr(mean), _ptr(invstd), _ptr(amax), _ptr(g),
                                             1 if ctx.training else 0, _ptr(gx), _ptr(gg), _ptr(gb), _ptr(ws), ws.numel(),
                                             _stream()), "pde_bn_pool_backward")
        return gx, (gg if ctx.has[0] else None), (gb if ctx.has[1] else None), None, None, None, None, None


def bn_pool_supported(x, bn) -> bool:
    """Whether ``bn_pool`` takes (x, bn): fp32 CUDA tensor (B,C,N,N), N a multiple of 4 up to 64, a BatchNorm2d with a
    fixed momentum (or no running statistics)."""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[2] == x.shape[3] and x.shape[2] % 4 == 0
            and 4 <= x.shape[2] <= 64 and x.shape[0] > 0 and not torch.is_autocast_enabled()
            and (bn.momentum is not None or not bn.track_running_stats)
            and (bn.training or bn.running_mean is not None))


def bn_pool(x, bn):
    """``cat([adaptive_avg_pool2d(bn(x), 4), adaptive_max_pool2d(bn(x), 4)], dim=1)`` for a ``torch.nn.BatchNorm2d``
    module ``bn`` (cifar10.py:346-353) without materialising ``bn(x)``: (B,C,N,N) -> (B,2C,4,4).  Updates the module's
    running statistics in training mode exactly as the module would."""
    training = bn.training or bn.running_mean is None
    rm = bn.running_mean if (bn.track_running_stats and bn.training) or not training else None
    rv = bn.running_var if (bn.track_running_stats and bn.training) or not training else None
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    return _BnPoolFn.apply(x, bn.weight, bn.bias, rm, rv, training, bn.momentum if bn.momentum is not None else 0.0, bn.eps)


# --------------------------------------------------------------------------- Ruthotto-Haber symmetric layer (MFMA)
_ACT_CODE = {"identity": 0, "relu": 1, "tanh": 2}


_sym_ws = {}


def _sym_workspace(B, D, dev):
    """Scratch for the partial tiles of the symmetric layer's split strip products: kept per (device, stream, width) —
    calls on one stream are ordered, calls on different streams get different buffers (and a captured graph keeps pointing
    at memory that stays allocated)."""
    n = L.load().pde_sym_layer_workspace_bytes(B, D)
    if n == 0:
        return None
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, D, n)
    ws = _sym_ws.get(key)
    if ws is None:
        if len(_sym_ws) > 64:
            _sym_ws.clear()
        ws = _sym_ws[key] = torch.empty(n, dtype=torch.uint8, device=dev)
    return ws


class _SymLayerFn(torch.autograd.Function):
    """out = base + scale * (act(BatchNorm1d(X K^T)) K) — cifar_2version.py:190-258 — on the fp32 matrix cores (pde_rh.hip)."""

    @staticmethod
    def forward(ctx, X, K, gamma, beta, base, running_mean, running_var, training, momentum, eps, scale, act):
        lib = L.load()
        _require_cuda(X, K, gamma, beta, base)
        B, D = X.shape
        Xf = X.to(torch.float32).contiguous()
        Kf = K.detach().to(torch.float32).contiguous()
        gm = gamma.detach().to(torch.float32).contiguous()
        bt = beta.detach().to(torch.float32).contiguous()
        bs = None if base is None else base.to(torch.float32).contiguous()
        dev = X.device
        P = torch.empty((B, D), dtype=torch.float32, device=dev)
        H = torch.empty_like(P)
        out = torch.empty_like(P)
        mean = torch.empty(D, dtype=torch.float32, device=dev)
        invstd = torch.empty_like(mean)
        with torch.cuda.device(dev):
            ws = _sym_workspace(B, D, dev)
            L.check(lib.pde_sym_layer_forward(B, D, act, 1 if training else 0, _ptr(Xf), _ptr(Kf), _ptr(gm), _ptr(bt),
                                              _ptr(running_mean), _ptr(running_var), float(momentum), float(eps), _ptr(bs),
                                              float(scale), _ptr(P), _ptr(H), _ptr(mean), _ptr(invstd), _ptr(out),
                                              _ptr(ws), 0 if ws is None else ws.numel(), _stream()),
                    "pde_sym_layer_forward")
        ctx.save_for_backward(Xf, Kf, gm, P, H, mean, invstd)
        ctx.cfg = (bool(training), float(scale), int(act), base is not None)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = L.load()
        Xf, Kf, gm, P, H, mean, invstd = ctx.saved_tensors
        training, scale, act, has_base = ctx.cfg
        B, D = Xf.shape
        g = gout.to(torch.float32).contiguous()
        dev = Xf.device
        dP = torch.empty_like(Xf)
        gX = torch.empty_like(Xf)
        gK = torch.empty_like(Kf)
        gg = torch.empty(D, dtype=torch.float32, device=dev)
        gb = torch.empty_like(gg)
        with torch.cuda.device(dev):
            ws = _sym_workspace(B, D, dev)
            L.check(lib.pde_sym_layer_backward(B, D, act, 1 if training else 0, _ptr(g), float(scale), _ptr(Xf), _ptr(Kf),
                                               _ptr(gm), _ptr(P), _ptr(H), _ptr(mean), _ptr(invstd), _ptr(dP), _ptr(gX),
                                               _ptr(gK), _ptr(gg), _ptr(gb), _ptr(ws), 0 if ws is None else ws.numel(),
                                               _stream()), "pde_sym_layer_backward")
        return gX, gK, gg, gb, (g if has_base else None), None, None, None, None, None, None, None


SYM_LAYER_MAX_ROWS = 128      # one strip workgroup owns all rows of its 16 features up to here (the reference trains at 64)


def sym_layer_supported(X, bn) -> bool:
    """Whether ``sym_layer`` takes this input: an fp32 CUDA batch whose feature count is a multiple of
    64, outside autocast, and a BatchNorm1d with affine parameters and a fixed momentum."""
    if not (X.is_cuda and X.dtype == torch.float32 and X.dim() >= 2 and X.shape[0] > 0) or torch.is_autocast_enabled():
        return False
    D = X[0].numel()
    if bn.weight is None or bn.bias is None or (bn.momentum is None and bn.track_running_stats):
        return False
    if not bn.training and bn.running_mean is None:
        pass                                              # eval without running statistics = batch statistics: supported
    B = X.shape[0]
    # Policy, not capability (``sym_layer`` itself takes any batch): above SYM_LAYER_MAX_ROWS rows the row-block kernels
    # are 1.6-1.9x behind rocBLAS, so the module-level callers keep plain torch there; a training-mode
    # batch of one row is refused by torch.nn.BatchNorm1d ("Expected more than 1 value per channel") and must stay so.
    if B > SYM_LAYER_MAX_ROWS or (B == 1 and (bn.training or bn.running_mean is None)):
        return False
    return bool(L.load().pde_sym_layer_supported(B, D))


def sym_layer(X, K, bn, activation: str = "relu", base=None, scale: float = -1.0):
    """``base + scale * (act(bn(X @ K.T)) @ K)`` for a dense (D, D) weight ``K`` and a ``torch.nn.BatchNorm1d`` ``bn``
    (cifar_2version.py:210-219; ``F_sym`` itself is ``base=None, scale=-1``; ParabolicBlock's step is ``base=Y, scale=-dt``,
    HamiltonianBlock's ``base=Y, scale=+dt`` on Z and ``base=Z, scale=+dt`` on Y).  X: (B, ...) flattened to (B, D); the result
    has X's shape.  Updates the module's running statistics in training mode exactly as the module would."""
    shape = X.shape
    B = shape[0]
    X2 = X.reshape(B, -1)
    training = bn.training or bn.running_mean is None
    track = bn.track_running_stats and bn.running_mean is not None
    rm = bn.running_mean if (track and bn.training) or not training else None
    rv = bn.running_var if (track and bn.training) or not training else None
    if bn.training and track and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    b2 = None if base is None else base.reshape(B, -1)
    mom = bn.momentum if bn.momentum is not None else 0.0
    H_ = L.host_ext()
    if H_ is not None and X2.is_cuda:
        # the native host path (csrc/host_ext.cpp): the same two calls of the C ABI from a C++ autograd node
        out = H_.sym(X2, K, bn.weight, bn.bias, b2, rm, rv, bool(training), float(mom), float(bn.eps), float(scale),
                     _ACT_CODE[activation])
    else:
        out = _SymLayerFn.apply(X2, K, bn.weight, bn.bias, b2, rm, rv, training, mom, bn.eps, scale, _ACT_CODE[activation])
    return out.view(shape)


# --------------------------------------------------------------------------- attention gates + weighted combination
class _GateCombineFn(torch.autograd.Function):
    """combined = sum_i w_i gate_i[b,c] y_i — cifar10.py:242 (x * attention_weights) and :277-280 in one pass; the
    backward in one pass too (scaled gradients of the y_i and the per-plane dots the gate/weight gradients need)."""

    @staticmethod
    def forward(ctx, weights, nl, *rest):
        lib = L.load()
        ys, gates = rest[:nl], rest[nl:]
        _require_cuda(weights, *rest)
        y0 = ys[0]
        dt = y0.dtype if y0.dtype in (torch.float32, torch.bfloat16) else torch.float32
        ys_c = [y.to(dt).contiguous() for y in ys]
        B, Cc = y0.shape[:2]
        HW = y0[0, 0].numel()
        g_c = [g.detach().to(torch.float32).reshape(B, Cc).contiguous() for g in gates]
        w_c = weights.detach().to(torch.float32).contiguous()
        out = torch.empty_like(ys_c[0])
        yp = (C.c_void_p * nl)(*[y.data_ptr() for y in ys_c])
        gp = (C.c_void_p * nl)(*[g.data_ptr() for g in g_c])
        with torch.cuda.device(y0.device):
            L.check(lib.pde_gate_combine_forward(nl, B, Cc, HW, _io_dtype(ys_c[0]), yp, gp, _ptr(w_c), _ptr(out), _stream()),
                    "pde_gate_combine_forward")
        ctx.save_for_backward(w_c, *ys_c, *g_c)
        ctx.nl = nl
        ctx.meta = (weights.dtype, [y.dtype for y in ys], [(g.dtype, g.shape) for g in gates])
        return out

    @staticmethod
    def backward(ctx, g):
        lib = L.load()
        nl = ctx.nl
        w_c, *rest = ctx.saved_tensors
        ys_c, g_c = rest[:nl], rest[nl:]
        B, Cc = ys_c[0].shape[:2]
        HW = ys_c[0][0, 0].numel()
        g = g.to(ys_c[0].dtype).contiguous()
        gys = [torch.empty_like(y) for y in ys_c]
        dots = [torch.empty((B, Cc), dtype=torch.float32, device=g.device) for _ in range(nl)]
        yp = (C.c_void_p * nl)(*[y.data_ptr() for y in ys_c])
        gp = (C.c_void_p * nl)(*[t.data_ptr() for t in g_c])
        gyp = (C.c_void_p * nl)(*[t.data_ptr() for t in gys])
        dp = (C.c_void_p * nl)(*[t.data_ptr() for t in dots])
        with torch.cuda.device(g.device):
            L.check(lib.pde_gate_combine_backward(nl, B, Cc, HW, _io_dtype(g), _ptr(g), yp, gp, _ptr(w_c), gyp, dp, _stream()),
                    "pde_gate_combine_backward")
        wdt, ydts, gmeta = ctx.meta
        gw = torch.stack([(g_c[i] * dots[i]).sum() for i in range(nl)]).to(wdt)
        ggates = [(w_c[i] * dots[i]).to(gmeta[i][0]).reshape(gmeta[i][1]) for i in range(nl)]
        return (gw, None, *[t.to(d) for t, d in zip(gys, ydts)], *ggates)


def gate_combine(ys, gates, weights):
    """``sum_i weights[i] * gates[i][:, :, None, None] * ys[i]`` in one pass over the tensors (cifar10.py:242,277-280)."""
    return _GateCombineFn.apply(weights, len(ys), *ys, *gates)


# --------------------------------------------------------------------------- explicit layers
class _Explicit5Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, alpha_base, channel_scaling, dt, eps, max_coeff, relax, num_steps):
        lib = L.load()
        _require_cuda(u, alpha_base, channel_scaling)
        if u.dtype not in (torch.float32, torch.bfloat16):
            u = u.float()
        u = u.contiguous()
        B, Cc, H, W = u.shape
        a = alpha_base.detach().to(torch.float32).contiguous()
        s = channel_scaling.detach().to(torch.float32).contiguous()
        out = torch.empty_like(u)
        need_grad = any(ctx.needs_input_grad[:3])
        fused = (H, W) in ((64, 64), (32, 32), (16, 16))         # planes that stay in registers over all steps
        # inputs of steps 2..num_steps: what the backward reads (and what chains the steps for generic plane sizes): fp32
        # whatever the tensors' type
        states = torch.empty((num_steps - 1,) + tuple(u.shape), dtype=torch.float32, device=u.device) \
            if num_steps > 1 and (need_grad or not fused) else None
        with torch.cuda.device(u.device):
            L.check(lib.pde_explicit5_forward(B, Cc, H, W, _io_dtype(u), _ptr(u), _ptr(a), _ptr(s), dt, eps, max_coeff,
                                              relax, num_steps, _ptr(states), _ptr(out), _stream()), "pde_explicit5_forward")
        ctx.save_for_backward(u, states if need_grad else None, a, s)
        ctx.cfg = (dt, eps, max_coeff, relax, num_steps)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = L.load()
        u, states, a, s = ctx.saved_tensors
        dt, eps, max_coeff, relax, num_steps = ctx.cfg
        B, Cc, H, W = u.shape
        gout = gout.to(u.dtype).contiguous()
        gu = torch.empty_like(u)
        ga, gs = torch.empty_like(a), torch.empty_like(s)
        ws = _workspace(lib.pde_explicit5_backward_workspace_bytes(B, Cc, H, W, _io_dtype(u), num_steps), u.device)
        with torch.cuda.device(u.device):
            L.check(lib.pde_explicit5_backward(B, Cc, H, W, _io_dtype(u), _ptr(u), _ptr(states), _ptr(gout), _ptr(a), _ptr(s),
                                               dt, eps, max_coeff, relax, num_steps, _ptr(gu), _ptr(ga), _ptr(gs), _ptr(ws),
                                               ws.numel(), _stream()), "pde_explicit5_backward")
        return gu, ga, gs, None, None, None, None, None


def explicit5_step(u, alpha_base, channel_scaling, dt=0.01, eps=1e-6, max_coeff=0.15, relax=0.1, num_steps=1):
    """``num_steps`` relaxed explicit 5-point steps in one call — tiny_imagenet.py:38-49,53-72."""
    if u.shape[0] == 0 or num_steps < 1:
        return _empty_passthrough(u, alpha_base, channel_scaling)
    return _Explicit5Fn.apply(u, alpha_base, channel_scaling, float(dt), float(eps), float(max_coeff), float(relax),
                              int(num_steps))


class _JacobiFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, a_row, b_col, nt):
        lib = L.load()
        _require_cuda(u, a_row, b_col)
        u = u.float().contiguous()
        B, H, W = u.shape
        a = a_row.detach().float().contiguous()
        b = b_col.detach().float().contiguous()
        out = torch.empty_like(u)
        with torch.cuda.device(u.device):
            L.check(lib.pde_jacobi_forward(B, H, W, nt, _ptr(u), _ptr(a), _ptr(b), _ptr(out), _stream()),
                    "pde_jacobi_forward")
        ctx.save_for_backward(u, a, b)
        ctx.nt = nt
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = L.load()
        u, a, b = ctx.saved_tensors
        B, H, W = u.shape
        gout = gout.float().contiguous()
        gu, ga, gb = torch.empty_like(u), torch.empty_like(a), torch.empty_like(b)
        ws = _workspace(lib.pde_jacobi_backward_workspace_bytes(B, H, W, ctx.nt), u.device)
        with torch.cuda.device(u.device):
            L.check(lib.pde_jacobi_backward(B, H, W, ctx.nt, _ptr(u), _ptr(gout), _ptr(a), _ptr(b), _ptr(gu), _ptr(ga),
                                            _ptr(gb), _ptr(ws), ws.numel(), _stream()), "pde_jacobi_backward")
        return gu, ga, gb, None


def jacobi_diffuse(u, a_row, b_col, nt: int):
    """emotion_recognition.py:82-97 on (B,H,W): reflect-pad once, ``nt`` Jacobi updates."""
    if u.shape[0] == 0:
        return _empty_passthrough(u, a_row, b_col)
    return _JacobiFn.apply(u, a_row, b_col, int(nt))


# --------------------------------------------------------------------------- SVHN skip connection
class _SkipBlendFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u0, u, skip_weight):
        lib = L.load()
        _require_cuda(u0, u, skip_weight)
        if u0.shape != u.shape:
            raise L.PdeError(f"shapes differ: {tuple(u0.shape)} vs {tuple(u.shape)}")
        dt = u.dtype if u.dtype in (torch.float32, torch.bfloat16) else torch.float32
        a, b = u0.to(dt).contiguous(), u.to(dt).contiguous()
        w = skip_weight.detach().to(torch.float32).reshape(1).contiguous()
        out = torch.empty_like(b)
        with torch.cuda.device(b.device):
            L.check(lib.pde_skip_blend_forward(b.numel(), _io_dtype(b), _ptr(a), _ptr(b), _ptr(w), _ptr(out), _stream()),
                    "pde_skip_blend_forward")
        ctx.save_for_backward(a, b, w)
        ctx.in_dtypes = (u0.dtype, u.dtype, skip_weight.dtype, skip_weight.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = L.load()
        a, b, w = ctx.saved_tensors
        g = g.to(b.dtype).contiguous()
        g_a, g_b = torch.empty_like(a), torch.empty_like(b)
        g_w = torch.empty(1, dtype=torch.float32, device=b.device)
        ws = _workspace(lib.pde_skip_blend_backward_workspace_bytes(b.numel()), b.device)
        with torch.cuda.device(b.device):
            L.check(lib.pde_skip_blend_backward(b.numel(), _io_dtype(b), _ptr(g), _ptr(a), _ptr(b), _ptr(w), _ptr(g_a), _ptr(g_b),
                                                _ptr(g_w), _ptr(ws), ws.numel(), _stream()), "pde_skip_blend_backward")
        d0, d1, dw, shw = ctx.in_dtypes
        return g_a.to(d0), g_b.to(d1), g_w.to(dw).reshape(shw)


def skip_blend(u0, u, skip_weight):
    """``sigmoid(skip_weight) * u0 + (1 - sigmoid(skip_weight)) * u`` — SVHN.py:73-74, one pass."""
    if u.numel() == 0:
        s = torch.sigmoid(skip_weight)
        return s * u0 + (1 - s) * u
    return _SkipBlendFn.apply(u0, u, skip_weight)


# --------------------------------------------------------------------------- timing
def timing_enable(on: bool = True):
    L.check(L.load().pde_timing_enable(int(on)), "pde_timing_enable")


def timing_read() -> Tuple[float, int, float, int]:
    """(sum of forward-kernel ms, launches, sum of backward-kernel ms, launches) since enable."""
    f, b = C.c_double(), C.c_double()
    nf, nb = C.c_int64(), C.c_int64()
    L.check(L.load().pde_timing_read(C.byref(f), C.byref(nf), C.byref(b), C.byref(nb)), "pde_timing_read")
    return f.value, nf.value, b.value, nb.value
