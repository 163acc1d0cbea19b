#!/usr/bin/env python3
"""Headline benchmark: forward+backward of the PDE diffusion layer (the hot path).

    python bench.py [--gpus N] [--steps K] [--warmup W]

N = 1 runs in-process.  For N > 1 the file runs as one rank per GPU over RCCL: either the driver launches it
under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or — called plainly as
``python bench.py --gpus N`` — it starts ``python -m torch.distributed.run --nproc-per-node N`` on itself as a
CHILD process before anything touches the GPU and exits with the child's code.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1]): cifar10.EnhancedDiffusionLayer
(size 32, channels 64, dt 1e-3, 10 Strang steps = 30 implicit sweeps), batch 512 per GPU, fp32,
synthetic N(0,1) input and upstream gradient, "trained-like" coefficients.  A step is one
forward + backward of the layer over one batch; with N > 1 the batch is sharded (weak scaling)
and the layer's parameter gradients are all-reduced over RCCL every step.
Primary line: channel mixing disabled (diffusion path only — SURVEY.md §8d declares it the
primary configuration; the C=64 mixing product is a GEMM outside the stencil path).  The
same workload WITH channel mixing is reported under "secondary", and the other BASELINE.json
configurations (SURVEY §8d cfg1, cfg3 at C=1 and C=32, cfg4 bf16, cfg5) under "configs", each timed the same
way (warm-up, K steps between synchronisations, max over ranks) with its algorithmic-bytes HBM fraction.
"""
import argparse
import contextlib
import io
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
BYTES_PER_ELEM = {"fwd": 8, "bwd": 12, "step": 20}   # fp32: u,y | gy,y,gu | total


def build_layer(C, N, steps, dev, rank, mixing):
    import cnn_with_pde_amd as P
    with contextlib.redirect_stdout(io.StringIO()):
        layer = P.EnhancedDiffusionLayer(N, C, dt=0.001, num_steps=steps, channel_mixing_enabled=mixing)
    g = torch.Generator().manual_seed(99)       # same parameters on every rank
    with torch.no_grad():                       # "trained-like" 
        for n, p in layer.named_parameters():
            if n in ("alpha_base", "beta_base"):
                p.mul_(1 + 0.1 * torch.randn(p.shape, generator=g))
            elif n in ("alpha_time_coeff", "beta_time_coeff"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    return layer.to(dev)


def run_steps(layer, u, gy, n, dist_on, flat, spans=None):
    """`spans` (multi-rank runs): receives one (before, after) event pair per step around the wait for the gradient
    all-reduce — what the stream waited for a collective that the backward's hooks had launched earlier."""
    for _ in range(n):
        if dist_on:
            flat.zero()                         # the gradients are views of the flat bucket: one launch clears them
        else:
            for p in layer.parameters():
                p.grad = None
        u.grad = None
        y = layer(u)
        y.backward(gy)
        if dist_on:
            # one flat bucket (1.06 MB, latency-bound over xGMI).  Headline: the collective was launched from inside
            # backward by the bucket's hooks (ReduceOp.AVG: no separate division), finish() only waits for it; the other
            # legs (some of their parameters receive no gradient) launch it here.
            if not flat._hooks:
                flat.start(average=True)
            if spans is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                flat.finish()
                e1.record()
                spans.append((e0, e1))
            else:
                flat.finish()


PRECONDITION_BATCH = 50    # set-up (see main): untimed batches of this many steps of the headline workload ...
PRECONDITION_MAX_S = 6.0   # ... until the batch time has stopped falling, or this many seconds


def precondition(layer, u, gy, dist_on, flat):
    """Untimed set-up: run the workload until its step time has stopped falling (three batches in a row within 1 % of the
    best seen), at most PRECONDITION_MAX_S seconds.  Every rank takes the same decision (the batch times are max-reduced)."""
    import torch.distributed as dist
    best, steady, hist = None, 0, []
    t_start = time.perf_counter()
    while True:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(layer, u, gy, PRECONDITION_BATCH, dist_on, flat)
        torch.cuda.synchronize()
        bt = (time.perf_counter() - t0) / PRECONDITION_BATCH * 1e3
        elapsed = time.perf_counter() - t_start
        if dist_on:
            t = torch.tensor([bt, elapsed], device=u.device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            bt, elapsed = float(t[0]), float(t[1])
        hist.append(bt)
        steady = steady + 1 if (best is not None and bt <= best * 1.01) else 0
        best = bt if best is None else min(best, bt)
        if (steady >= 3 and len(hist) >= 4) or elapsed > PRECONDITION_MAX_S:
            break
    return {"steps": len(hist) * PRECONDITION_BATCH, "seconds": round(elapsed, 2),
            "ms_per_step_first_batch": round(hist[0], 4), "ms_per_step_last_batch": round(hist[-1], 4)}
LAST_DIAG = {}        # per-rank figures of the latest timed() call on a multi-rank run (rank 0 puts them in the JSON line)


def timed(layer, u, gy, steps, warmup, dist_on, flat):
    import torch.distributed as dist
    run_steps(layer, u, gy, warmup, dist_on, flat)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    spans = [] if dist_on else None
    run_steps(layer, u, gy, steps, dist_on, flat, spans)
    torch.cuda.synchronize()
    own = time.perf_counter() - t0               # this rank alone, before it waits for the others
    if dist_on:
        dist.barrier()
    dt = time.perf_counter() - t0
    LAST_DIAG.clear()
    if dist_on:
        t = torch.tensor([dt], device=u.device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # what a first multi-GPU run needs in order to be read: every rank's own step time and the part of it the
        # stream spent waiting for the gradient all-reduce (the collective is launched from inside backward)
        exposed = sum(a.elapsed_time(b) for a, b in spans) / max(len(spans), 1)
        mine = torch.tensor([own / steps * 1e3, exposed], device=u.device, dtype=torch.float64)
        allv = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(allv, mine)
        per = [float(v[0]) for v in allv]
        exp = [float(v[1]) for v in allv]
        LAST_DIAG.update({"rank_ms_per_step": {"min": min(per), "max": max(per), "all": [round(x, 4) for x in per]},
                          "allreduce_exposed_ms": {"mean": sum(exp) / len(exp), "max": max(exp),
                                                   "note": "per step: stream time between the end of backward and the "
                                                           "completion of the gradient all-reduce its hooks launched "
                                                           "(events around GradBucket.finish())"}})
    return dt


def graph_replay_ms(layer, u, gy, reps):
    """Forward + backward of ``layer`` on (u, gy) captured once in a hipGraph (cnn_with_pde_amd.graphs: checkpoint plan
    frozen from the current coefficients, so the library's calls are launches only) and replayed ``reps`` times."""
    import cnn_with_pde_amd as P
    layer.freeze_checkpoint_plan(u)
    params = [p_ for p_ in layer.parameters() if p_.requires_grad]
    step = P.GraphedStep(lambda: torch.autograd.grad(layer(u), [u] + params, gy))
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def cpu_baseline(C, N, steps, sample_B):
    """The oracle in reference-faithful mode (per-unknown Python loop of torch ops, autograd
    backward — what the reference does) on this box's host cores, on a FIXED sample of B = 32 of the
    same workload: one forward+backward, seconds and thread count reported.  (Rounds 1-2
    sized the sample from a probe; the faithful oracle's autograd graph then fell over a memory cliff
    on some hosts and the figure moved 8x between rounds.)"""
    from oracle import pde_oracle as O
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))             # the GPU box's CPU share for one GPU is 16
    torch.set_num_threads(cores)
    spec = O.AdiSpec(N, C, 0.001, 1.0, 1.0, steps, "strang", False, 10.0, "none", False)
    g = torch.Generator().manual_seed(5)
    params = O.adi_init_params(spec, "cifar10", gen=g)
    params.pop("channel_mixing", None)

    def once(nb):
        u = torch.randn(nb, C, N, N, generator=g)
        gy = torch.randn(nb, C, N, N, generator=g)
        t0 = time.perf_counter()
        O.value_and_grads(lambda a, p: O.adi_forward(a, p, spec), u, params, gy)
        return time.perf_counter() - t0

    nb = int(sample_B)
    print(f"[bench] cpu_baseline: oracle, B={nb}, {cores} threads ...", file=sys.stderr, flush=True)
    dt = once(nb)
    print(f"[bench] cpu_baseline: {dt:.1f} s", file=sys.stderr, flush=True)
    out = {"value": nb / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
           "seconds": dt, "threads": cores,
           "sample": f"fixed B={nb} of the same (C={C},{N}x{N},{steps} steps) workload, one fwd+bwd, "
                     f"oracle/pde_oracle.py in reference-faithful mode ({dt:.1f} s on {cores} threads)"}
    try:                                        # also: the C restatement with closed-form gradients (OpenMP)
        from oracle import c_oracle as CO
        if CO.available():
            os.environ.setdefault("OMP_NUM_THREADS", str(cores))
            nb2 = 64
            u = torch.randn(nb2, C, N, N, generator=g)
            gy = torch.randn(nb2, C, N, N, generator=g)
            t0 = time.perf_counter()
            CO.adi_value_and_grads(u, params, gy, spec)
            dt2 = time.perf_counter() - t0
            out["c_restatement"] = {"value": nb2 / dt2 / 1e6, "unit": "Msamples/s",
                                    "sample": f"B={nb2}, oracle/pde_oracle_c.c fp32, OpenMP {cores} threads, {dt2:.2f} s"}
    except Exception as e:                      # the checker's availability must not break the bench
        out["c_restatement"] = {"error": repr(e)}
    return out


def pmc_extra(work):
    """HBM bytes per forward+backward of a workload from the committed counter passes (FETCH_SIZE doubled + WRITE_SIZE over
    ALL this library's launches of a step: profiles/pmc_traffic.json, tools/pmc_round.sh), or None"""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.isfile(path):
        return None
    with open(path) as f:
        pj = json.load(f)
    if work + "_bytes_per_step" not in pj:
        return None
    return {"hbm_bytes_per_step": pj[work + "_bytes_per_step"], "taken_at_commit": pj.get("commit"),
            "largest_kernels": pj.get(work + "_kernels")}


def config_legs(dev, rank, world, dist_on, quick):
    """The other BASELINE.json configurations, per-GPU batch fixed (weak scaling), each timed like
    the headline: warm-up, K steps between barrier+synchronize, max over ranks, gradients all-reduced when N > 1.
    bytes/element: 20 fp32, 10 bf16 (SURVEY §8d algorithmic bytes)."""
    import cnn_with_pde_amd as P
    legs = {}

    def leg(name, layer, shape, dtype, bpe, steps, desc, graph=False):
        layer = layer.to(dev)
        warm = 10
        if graph:                                  # the launch-bound legs: the host path (allocator, ctypes, autograd) and the
            warm, steps = 100, 10 * steps          # clocks take ~100 calls to settle (0.27 -> 0.15 ms on the same box)
        g = torch.Generator().manual_seed(4321 + rank)
        u = torch.randn(*shape, generator=g).to(dtype).to(dev).requires_grad_(True)
        gy = torch.randn(*shape, generator=g).to(dtype).to(dev)
        flat = P.GradBucket(layer.parameters(), grads_as_views=True) if dist_on else None
        dt = timed(layer, u, gy, steps, warm, dist_on, flat)
        ms = dt / steps * 1e3
        gbs = u.numel() * bpe / (dt / steps) / 1e9            # per GPU
        if rank == 0:
            legs[name] = {"workload": desc, "per_gpu_shape": list(shape), "dtype": str(dtype).replace("torch.", ""),
                          "steps": steps, "ms_per_step": ms, "value": shape[0] * world / (dt / steps) / 1e6,
                          "unit": "Msamples/s", "algorithmic_GBps_per_gpu": gbs, "frac": gbs / HBM_PEAK_GBS,
                          "bytes_per_element": bpe}
        if graph and not dist_on:
            # host-bound shapes: the same forward+backward captured once in a hipGraph and replayed (cnn_with_pde_amd.graphs)
            try:
                gms = graph_replay_ms(layer, u, gy, 4 * steps)
                if rank == 0:
                    legs[name]["ms_per_step_hipgraph_replay"] = gms
                    legs[name]["frac_hipgraph_replay"] = u.numel() * bpe / (gms * 1e-3) / 1e9 / HBM_PEAK_GBS
            except Exception as e:                             # reported, never fatal for the bench line
                if rank == 0:
                    legs[name]["ms_per_step_hipgraph_replay"] = "capture failed: %s" % (str(e)[:120],)
        del layer, u, gy
        torch.cuda.empty_cache()

    k = 5 if quick else 20
    with contextlib.redirect_stdout(io.StringIO()):
        mn = P.MnistDiffusionLayer()
        fa = P.FashionDiffusionLayer()
        c32 = P.SvhnDiffusionLayer(28, 32, dt=0.3, num_steps=4)
        c4 = P.SvhnDiffusionLayer(32, 128, num_steps=20)
        c5 = P.ImprovedDiffusionLayer(64, 64)
    gp = torch.Generator().manual_seed(7)
    with torch.no_grad():
        # cfg3 at 32 channels: fashion semantics broadcast per channel = the SVHN layer with coupling I and the
        # skip weight at -40 (sigmoid ~ 4e-18: the output is the diffused branch) — SURVEY §8d
        c32.alpha_base.fill_(1.8); c32.beta_base.fill_(1.8); c32.alpha_time_coeff.zero_(); c32.beta_time_coeff.zero_()
        c32.channel_coupling.copy_(torch.eye(32)); c32.skip_weight.fill_(-40.0)
        c4.channel_coupling.copy_(torch.eye(128) + 0.01 * torch.randn(128, 128, generator=gp))
    leg("cfg1", mn, (64, 1, 28, 28), torch.float32, 20, k, "mnist_test.DiffusionLayer(), batch 64 (BASELINE configs[0])", graph=True)
    leg("cfg3_c1", fa, (4096, 1, 28, 28), torch.float32, 20, k,
        "fashion_mnist.DiffusionLayer() literal C=1, batch 4096/GPU (BASELINE configs[2])", graph=True)
    leg("cfg3_c32", c32, (512, 32, 28, 28), torch.float32, 20, k,
        "SVHN.DiffusionLayer(28,32,dt=0.3,num_steps=4), fashion coefficients, coupling each step, batch 512/GPU")
    leg("cfg4_bf16", c4, (512, 128, 32, 32), torch.bfloat16, 10, max(3, k // 4),
        "SVHN.DiffusionLayer(32,128,num_steps=20) bf16 tensors, 60 sweeps + 20 couplings + skip, batch 512/GPU (BASELINE configs[3])")
    leg("cfg5", c5, (256, 64, 64, 64), torch.float32, 20, k,
        "tiny_imagenet.ImprovedDiffusionLayer(64,64) explicit 5-point step, batch 256/GPU (BASELINE configs[4])")
    if rank == 0 and "cfg5" in legs:
        # counter traffic of the two explicit kernels (separate FETCH_SIZE / WRITE_SIZE passes, profiles/pmc_traffic.json)
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pj = json.load(f)
            if "explicit5_fwd_wave_bytes_per_launch" in pj:
                legs["cfg5"]["traffic"] = {"explicit5_fwd_wave": pj["explicit5_fwd_wave_bytes_per_launch"],
                                           "explicit5_bwd_wave": pj["explicit5_bwd_wave_bytes_per_launch"],
                                           "algorithmic": {"forward": 256 * 64 * 64 * 64 * 8, "backward": 256 * 64 * 64 * 64 * 12},
                                           "taken_at_commit": pj.get("commit"), "unit": "bytes per launch"}
        except (OSError, ValueError):
            pass

    # one application of cifar_2version.SymmetricLayer at the reference's own size (3 x 32 x 32: K is 3072^2)
    # and batch (128), forward + backward, on the fp32 matrix cores (pde_rh.hip); the same module in plain torch (rocBLAS)
    # beside it.  MFMA-bound: 6 products of 2*B*D^2 flop (2 forward, 4 backward).
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            sl = P.SymmetricLayer(3, 32).to(dev).train()
        g = torch.Generator().manual_seed(77 + rank)
        xs = torch.randn(128, 3, 32, 32, generator=g).to(dev).requires_grad_(True)
        gs = torch.randn(128, 3, 32, 32, generator=g).to(dev)

        def sl_ms(fused, reps):
            sl.fused = fused
            def one():
                for p_ in sl.parameters():
                    p_.grad = None
                xs.grad = None
                sl(xs).backward(gs)
            for _ in range(5):
                one()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                one()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3
        ms_f, ms_t = sl_ms(True, 2 * k), sl_ms(False, 2 * k)
        # the reference's own training batch is 64 (cifar_2version.py:476)
        xs = torch.randn(64, 3, 32, 32, generator=g).to(dev).requires_grad_(True)
        gs = torch.randn(64, 3, 32, 32, generator=g).to(dev)
        ms_f64, ms_t64 = sl_ms(True, 2 * k), sl_ms(False, 2 * k)
        flop = 6 * 2.0 * 128 * 3072 * 3072
        if rank == 0:
            legs["rh_symmetric"] = {"workload": "cifar_2version.SymmetricLayer(3, 32): -act(BN(Y K^T)) K, K 3072 x 3072 fp32, batch 128 "
                                                "(twice the reference's training batch of 64, cifar_2version.py:476: `batch_64` beside "
                                                "it), training mode, forward + backward (cifar_2version.py:190-220)",
                                    "batch_64": {"ms_per_step": ms_f64, "ms_per_step_plain_torch_rocblas": ms_t64},
                                    "ms_per_step": ms_f, "ms_per_step_plain_torch_rocblas": ms_t, "value": 128 / ms_f / 1e3,
                                    "unit": "Msamples/s", "dtype": "f32",
                                    "roofline": {"bound": "mfma", "achieved": flop / (ms_f * 1e-3) / 1e12, "peak": 157.3,
                                                 "unit": "TFLOP/s", "frac": flop / (ms_f * 1e-3) / 1e12 / 157.3,
                                                 "flop_per_step": flop,
                                                 "note": "fp32-input MFMA (v_mfma_f32_32x32x2_f32) peak, MI355X_MICROARCH.md; whole "
                                                         "step incl. launches and the autograd path"}}
        del sl, xs, gs
    except Exception as e:                                   # reported, never fatal for the bench line
        if rank == 0:
            legs["rh_symmetric"] = {"error": repr(e)[:200]}

    # the reference's OWN shapes (C = 3): cifar10.MultiScaleExtractor's three PDE layers on one 128-sample batch,
    # one launch per pass; host-launch-bound, reported as time per forward+backward
    with contextlib.redirect_stdout(io.StringIO()):
        trio = [P.EnhancedDiffusionLayer(32, 3, dt=0.001, num_steps=5, dx=1.0, dy=1.0).to(dev),
                P.EnhancedDiffusionLayer(32, 3, dt=0.002, num_steps=8, dx=2.0, dy=2.0).to(dev),
                P.EnhancedDiffusionLayer(32, 3, dt=0.005, num_steps=4, dx=1.5, dy=1.5).to(dev)]
    g = torch.Generator().manual_seed(99 + rank)
    x = torch.randn(128, 3, 32, 32, generator=g).to(dev).requires_grad_(True)
    gx = torch.randn(128, 3, 32, 32, generator=g).to(dev)
    w = torch.full((3,), 1.0 / 3, device=dev, requires_grad=True)

    def trio_step(fused):
        for ly in trio:
            for p_ in ly.parameters():
                p_.grad = None
        x.grad = None
        if fused:
            out, _ = P.diffuse_shared_input(trio, x, w)
        else:
            out = sum(wi * ly(x) for wi, ly in zip(w, trio))
        out.backward(gx)
    res = {}
    for fused in (True, False):
        for _ in range(100):                               # launch-bound: see leg()
            trio_step(fused)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10 * k):
            trio_step(fused)
        torch.cuda.synchronize()
        res[fused] = (time.perf_counter() - t0) / (10 * k) * 1e3
    # the same step replayed from a hipGraph (cnn_with_pde_amd.graphs): explicit checkpoint plans, launches only
    graph_ms = None
    try:
        if dist_on:
            raise RuntimeError("single-GPU runs only")
        for ly in trio:
            ly.freeze_checkpoint_plan(x)
        params = [p_ for ly in trio for p_ in ly.parameters()]
        step = P.GraphedStep(lambda: torch.autograd.grad(P.diffuse_shared_input(trio, x, w)[0], [x, w] + params, gx))
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6 * k):
            step()
        torch.cuda.synchronize()
        graph_ms = (time.perf_counter() - t0) / (6 * k) * 1e3
    except Exception as e:                                   # reported, never fatal for the bench line
        graph_ms = "capture failed: %s" % (str(e)[:120],)
    if rank == 0:
        legs["cifar10_trio_c3"] = {"workload": "the three EnhancedDiffusionLayers of cifar10.MultiScaleExtractor (C=3, 5/8/4 steps) on one "
                                               "(128,3,32,32) batch + weighted sum, forward+backward (reference shapes, cifar10.py:251-280)",
                                   "ms_per_step": res[True], "ms_per_step_one_call_per_layer": res[False],
                                   "ms_per_step_hipgraph_replay": graph_ms,
                                   "value": 128 / res[True] / 1e3, "unit": "Msamples/s",
                                   "note": "one launch per pass for all three layers (pde_adi_multi_*); the eager figures are bound by the host's "
                                           "launch path, ms_per_step_hipgraph_replay is the same forward+backward captured once "
                                           "(cnn_with_pde_amd.graphs.GraphedStep, checkpoint plans frozen) and replayed"}
    return legs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="per-GPU batch")
    ap.add_argument("--channels", type=int, default=64)
    ap.add_argument("--size", type=int, default=32)
    ap.add_argument("--num-steps", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-config legs (cfg1, cfg3, cfg4, cfg5)")
    ap.add_argument("--cpu-sample", type=int, default=32, help="samples of the CPU baseline (SURVEY §8d: fixed B = 32)")
    ap.add_argument("--rehearse-one-device", action="store_true",
                    help="developer rehearsal of the multi-rank path on a ONE-GPU box: every rank uses cuda:0 and the "
                         "collectives run over gloo (RCCL refuses two ranks on one device); timings are meaningless")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
    if a.gpus != world and dist_on:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and not dist_on:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, as a child process, BEFORE any GPU call
        # in this one (never a re-exec), and hand its stdout (rank 0's one JSON line) and exit code through
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if a.rehearse_one_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rccl_ws = 1
    if dist_on:
        import torch.distributed as dist
        if a.rehearse_one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        rccl_ws = dist.get_world_size()

    import cnn_with_pde_amd as P
    B, C, N, steps = a.batch, a.channels, a.size, a.num_steps
    # the CPU path first, before the GPU legs hold host memory (fixed B = 32)
    cpu_base = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu_base = cpu_baseline(C, N, steps, a.cpu_sample)
    g = torch.Generator().manual_seed(1234 + rank)
    u = torch.randn(B, C, N, N, generator=g).to(dev).requires_grad_(True)
    gy = torch.randn(B, C, N, N, generator=g).to(dev)
    layer = build_layer(C, N, steps, dev, rank, mixing=False)
    layer.channel_mixing.requires_grad_(False)          # unused when mixing is disabled
    flat = None
    if dist_on:
        flat = P.GradBucket(layer.parameters(), grads_as_views=True)
        flat.attach_hooks(average=True)         # the all-reduce leaves from inside backward, behind the last gradient

    # The timed loop is a training loop: the layer plans its backward's checkpoints from the PREVIOUS step's coefficient
    # maxima ("lagged", layers.py — the policy meant for loops whose parameters move by optimiser steps).  The default
    # ("auto": this call's own maxima) makes the backward wait for the forward's factorisation kernel, so the host can run
    # at most one forward ahead of the device; a host that is late by more than that now and then (seen on 2 of 13
    # boxes: +0.1 ms per step) shows in the step time.  The same loop under "auto" is timed beside it (`eager_auto_policy`).
    layer.checkpoint_policy = "lagged"
    # Set-up, before the W warm-up steps and the K timed ones: the device and the allocator in their steady state.  A
    # chip that has been idle needs this load for a while before it holds its clocks (with 5 warm-up steps the first
    # timed leg read 0.499 ms per step on a box whose steady state is 0.456; a freshly leased box was still falling
    # after 150 steps), and torch's caching allocator takes its 134 MB blocks from the driver during the first steps.
    precond = precondition(layer, u, gy, dist_on, flat)
    dt = timed(layer, u, gy, a.steps, a.warmup, dist_on, flat)
    headline_diag = dict(LAST_DIAG)
    layer.checkpoint_policy = "auto"
    dt_auto = timed(layer, u, gy, min(a.steps, 30), 3, dist_on, flat)
    auto_ms = dt_auto / min(a.steps, 30) * 1e3
    layer.checkpoint_policy = "lagged"
    ms_step = dt / a.steps * 1e3
    samples_s = B * world * a.steps / dt
    elems = B * C * N * N

    # per-kernel device time (HIP events on the launch stream, inside the library)
    P.timing_enable(True)
    run_steps(layer, u, gy, min(a.steps, 20), dist_on, flat)
    f_ms, f_n, b_ms, b_n = P.timing_read()
    P.timing_enable(False)
    fwd_ms, bwd_ms = f_ms / max(f_n, 1), b_ms / max(b_n, 1)

    # device-copy ceiling measured on this box (report against the spec peak and against this)
    src = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    for _ in range(3):
        dst.copy_(src)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        dst.copy_(src)
    torch.cuda.synchronize()
    copy_gbs = 20 * 2 * src.numel() / (time.perf_counter() - t0) / 1e9
    del src, dst

    # the same step replayed from a hipGraph (extra information: `value` above is the eager autograd path)
    graph_ms = None
    if not dist_on:
        try:
            graph_ms = graph_replay_ms(layer, u, gy, a.steps)
            layer.checkpoint_policy = "auto"
        except Exception as e:
            graph_ms = "capture failed: %s" % (str(e)[:120],)

    out = None
    # which kernel the backward of this schedule is (the hand-scheduled assembly kernel since round 4: include/pdecnn.h)
    bwd_kernel = "adi_bwd_kernel"
    try:
        import ctypes
        import cnn_with_pde_amd.functional as F_
        import cnn_with_pde_amd._lib as L_
        sw_ = [s_ for st_ in P.adi_schedule(layer.dt, layer.dx, layer.dy, steps) for s_ in st_]
        d_ = F_._build_desc(B, C, N, L_.PDE_IO_F32, sw_, False, 10.0, 1e-6)
        if L_.load().pde_adi_backward_kernel(ctypes.byref(d_), 0) == 1:
            bwd_kernel = "adi_bwd_asm_n32_w" + os.environ.get("PDE_ASM_VARIANT", "8b")
    except Exception:
        pass
    if rank == 0:
        pmc, valu, pmc_src = None, None, None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.isfile(pmc_path) and (B, C, N, steps) == (512, 64, 32, 10):     # counters were taken on this workload
            with open(pmc_path) as f:
                pj = json.load(f)
            pmc = pj.get("adi_bwd_kernel_bytes_per_launch") if pj.get("bwd_kernel", "adi_bwd_kernel").startswith(bwd_kernel[:11]) else None
            valu = pj.get("sq_insts_valu_per_launch")
            pmc_src = {"file": "profiles/pmc_traffic.json", "taken_at_commit": pj.get("commit"), "passes": pj.get("source")}
        ach = elems * BYTES_PER_ELEM["bwd"] / (bwd_ms * 1e-3) / 1e9
        out = {
            "metric": "PDE-layer fwd+bwd Msamples/s", "value": samples_s / 1e6, "unit": "Msamples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cifar10.EnhancedDiffusionLayer(size={N}, channels={C}, num_steps={steps}) "
                                   f"fwd+bwd, {3 * steps} implicit sweeps, batch {B}/GPU, channel mixing disabled "
                                   "(BASELINE configs[1], SURVEY §8d cfg2 primary)",
                       "global_batch": B * world, "parallelism": f"dp{world}",
                       "grad_allreduce_bytes": (flat.nbytes() if flat is not None else 0),
                       "rccl_world_size": rccl_ws},
            "preconditioning": dict(precond, note=f"untimed batches of {PRECONDITION_BATCH} steps of this workload during set-up, before "
                                                  f"the {a.warmup} warm-up steps, until the batch time has stopped falling (device "
                                                  "clocks and allocator pools in their steady state)"),
            "checkpoint_policy": "lagged (the backward's checkpoint plan comes from the previous step's coefficient maxima: no "
                                 "host wait inside the step; both plans are 'no checkpoints' on this workload)",
            "eager_auto_policy": {"ms_per_step": auto_ms, "value": B * world / (auto_ms * 1e-3) / 1e6, "unit": "Msamples/s",
                                  "note": "the same eager loop under the default policy (the backward waits for this call's "
                                          "own maxima); differs from `value` only when the host is late"},
            "hipgraph_replay": None if graph_ms is None else {
                "ms_per_step": graph_ms,
                "value": (B * world / (graph_ms * 1e-3) / 1e6) if isinstance(graph_ms, float) else None, "unit": "Msamples/s",
                "note": "the same forward+backward captured once (cnn_with_pde_amd.graphs.GraphedStep, checkpoint plan "
                        "frozen) and replayed: no Python/autograd/ctypes per step, but the replay also copies the gradients "
                        "(134 MB for the input's) into the step's static output buffers; `value` above is the eager path"},
            "roofline": {"bound": "hbm", "kernel": bwd_kernel, "achieved": ach, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": pmc, "traffic_source": pmc_src,
                         "copy_ceiling_measured": copy_gbs,
                         "algorithmic_bytes_per_launch": elems * BYTES_PER_ELEM["bwd"], "avg_launch_ms": bwd_ms},
            "roofline_fwd": {"bound": "hbm", "kernel": "adi_fwd_kernel",
                             "achieved": elems * BYTES_PER_ELEM["fwd"] / (fwd_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "avg_launch_ms": fwd_ms},
            "step_hbm": {"achieved": samples_s / world * C * N * N * BYTES_PER_ELEM["step"] / 1e9, "unit": "GB/s",
                         "frac": samples_s / world * C * N * N * BYTES_PER_ELEM["step"] / 1e9 / HBM_PEAK_GBS,
                         "note": "per GPU: samples/s x 20 B/element (whole step incl. launch gaps and small kernels)"},
        }
        if valu:
            # the second ceiling: VALU issue.  A SIMD takes one wave64 fp32 instruction per ~2 cycles from
            # two or more waves (tools/ubench/sweep_pk.hip: 2.3 with two, 1.6 with three), one wave issues at most one
            # per ~4.3 cycles; the backward has two waves per SIMD (246 VGPRs), the forward four.  1024 SIMDs at 2.4 GHz.
            def floors(n_inst, waves_per_simd):
                simd = n_inst * 2.0 / 1024 / 2.4e9 * 1e3
                per_wave = n_inst * 4.3 / waves_per_simd / 1024 / 2.4e9 * 1e3
                return simd, max(simd, per_wave)
            sb, wb = floors(valu["adi_bwd_kernel"], 2)
            sf, wf = floors(valu["adi_fwd_kernel"], 4)
            out["valu_issue"] = {"note": "SQ_INSTS_VALU per launch (rocprofv3 --pmc, profiles/) at the SIMD's rate (2 cycles per "
                                         "instruction) and at the per-wave issue limit (4.3 cycles, waves per SIMD: backward 2, "
                                         "forward 4); frac = floor / measured launch time",
                                 "adi_bwd_kernel": {"floor_ms_simd_rate": sb, "frac_simd_rate": sb / bwd_ms,
                                                    "floor_ms_per_wave_limit": wb, "frac_per_wave_limit": wb / bwd_ms},
                                 "adi_fwd_kernel": {"floor_ms_simd_rate": sf, "frac_simd_rate": sf / fwd_ms,
                                                    "floor_ms_per_wave_limit": wf, "frac_per_wave_limit": wf / fwd_ms}}

    if dist_on:
        # strong scaling beside the weak line: the SAME global batch of `--batch` samples split over the ranks
        lo, hi = P.shard_range(B, rank, world)
        us = u.detach()[: hi - lo].clone().requires_grad_(True)
        gs = gy[: hi - lo].clone()
        dts = timed(layer, us, gs, a.steps, a.warmup, dist_on, flat)
        if rank == 0:
            out["dist"] = headline_diag
            out["strong"] = {"dist": dict(LAST_DIAG),"scaling": "strong", "global_batch": B, "per_gpu_batch": hi - lo, "ms_per_step": dts / a.steps * 1e3,
                             "value": B * a.steps / dts / 1e6, "unit": "Msamples/s",
                             "note": "same layer, the global batch fixed at --batch and sharded over the ranks (contiguous "
                                     "shards, gradient all-reduce every step); `value` above is the weak-scaling line"}
        del us, gs

    if not a.no_secondary:
        layer2 = build_layer(C, N, steps, dev, rank, mixing=True)
        flat2 = P.GradBucket(layer2.parameters(), grads_as_views=True) if dist_on else None
        k2 = max(3, a.steps // 5)
        dt2 = timed(layer2, u, gy, k2, 2, dist_on, flat2)
        if rank == 0:
            out["secondary"] = {"config": "same workload with the C x C channel mixing before every step "
                                          "(cifar10.py:91): one factorisation, the forward in ONE launch (a workgroup owns all 64 channels of a "
                                          "sample, the operator through LDS as three-piece bf16 products on the bf16 MFMA, fp32-accurate); "
                                          "backward per step: 3-sweep adjoint launch + fused mixing-gradient kernel (three-piece bf16 "
                                          "products as well), gradients accumulated on the device",
                                "value": B * world * k2 / dt2 / 1e6, "unit": "Msamples/s",
                                "ms_per_step": dt2 / k2 * 1e3,
                                "algorithmic_bytes_per_step": B * C * N * N * BYTES_PER_ELEM["step"]}
            tr = pmc_extra("secondary")
            if tr and (B, C, N, steps) == (512, 64, 32, 10):
                out["secondary"]["traffic"] = tr
        del layer2

    if not a.no_configs:
        del u, gy, layer
        torch.cuda.empty_cache()
        legs = config_legs(dev, rank, world, dist_on, quick=a.steps < 20)
        if rank == 0:
            tr = pmc_extra("cfg4")
            if tr and isinstance(legs.get("cfg4_bf16"), dict):
                legs["cfg4_bf16"]["traffic"] = tr
            out["configs"] = legs

    if rank == 0:
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
        print(json.dumps(out))
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
