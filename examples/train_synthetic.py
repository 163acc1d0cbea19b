#!/usr/bin/env python3
"""End-to-end training step around the MI355X PDE layer, on synthetic data.

Counterpart of the reference's training entry points (model shape of mnist_test.py:221-235 /
the optimiser recipe of mnist_test.py:282-306: AdamW, cosine schedule, label smoothing, grad clipping),
with the diffusion layer taken from this package and data parallelism over RCCL:

    python examples/train_synthetic.py --variant mnist --steps 200        # single GPU, fp32: the step is ONE hipGraph
    python examples/train_synthetic.py --variant mnist --steps 200 --eager  # the same, one eager autograd step at a time
    python examples/train_synthetic.py --variant cifar10_noconv --amp          # cifar10.py:318-361 under fp16 autocast
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        examples/train_synthetic.py --variant svhn_model --steps 200

Variants: ``mnist`` / ``cifar10`` are small stand-ins; ``mnist_model``, ``fashion_model``, ``svhn_model``,
``cifar10_noconv``, ``tiny_model`` and ``emotion_model`` are the counterparts of the reference's own models
(cnn_with_pde_amd.models: mnist_test.py:223-237, fashion_mnist.py:200-224, SVHN.py:234-270, cifar10.py:318-361,
tiny_imagenet.py:237-305, emotion_recognition.py:170-195).  ``--amp`` runs the step under fp16 autocast with a
GradScaler, as cifar10.py:440,458-467 does.

There is no dataset on the box: every rank draws its shard of a fixed synthetic classification task
(one smooth random template per class plus noise), so the loss has something to learn and the run is
reproducible.  One flat bucket (cnn_with_pde_amd.GradBucket) all-reduces every gradient of the model.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cnn_with_pde_amd as P  # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):      # the layers print the reference's ctor banner
        return fn(*a, **k)


class MnistLike(nn.Module):
    """diff -> flatten -> dropout -> fc -> relu -> dropout -> fc, as mnist_test.py:221-235."""
    def __init__(self, size=28, classes=10):
        super().__init__()
        self.diff = quiet(P.MnistDiffusionLayer, size)
        self.dropout = nn.Dropout(0.1)
        self.fc1 = nn.Linear(size * size, 256)
        self.fc2 = nn.Linear(256, classes)

    def forward(self, x):
        x = self.diff(x).reshape(x.size(0), -1)
        x = F.relu(self.fc1(self.dropout(x)))
        return self.fc2(self.dropout(x))


class Cifar10Like(nn.Module):
    """A stem to `channels` feature maps, the multi-channel implicit layer, pooled linear head."""
    def __init__(self, size=32, channels=16, classes=10, steps=4):
        super().__init__()
        self.stem = nn.Conv2d(3, channels, 3, padding=1)
        self.pde = quiet(P.EnhancedDiffusionLayer, size, channels, dt=0.01, num_steps=steps)
        self.head = nn.Linear(channels * 16, classes)

    def forward(self, x):
        x = self.pde(F.relu(self.stem(x)))
        return self.head(F.adaptive_avg_pool2d(x, 4).flatten(1))


#: variant -> (model factory, (channels, size), classes)
VARIANTS = {
    "mnist": (lambda: MnistLike(), (1, 28), 10),
    "cifar10": (lambda: Cifar10Like(), (3, 32), 10),
    "mnist_model": (lambda: quiet(P.MnistPDEClassifier), (1, 28), 10),
    "fashion_model": (lambda: quiet(P.FashionPDEClassifier), (1, 28), 10),
    "svhn_model": (lambda: quiet(P.SvhnPDEClassifier), (3, 32), 10),
    "cifar10_noconv": (lambda: quiet(P.CIFAR10PDENoConv), (3, 32), 10),
    "tiny_model": (lambda: quiet(P.TinyImageNetClassifier, num_classes=20), (3, 64), 20),
    "emotion_model": (lambda: quiet(P.EmotionDiffusionClassifier), (1, 48), 7),
}


def synthetic_task(variant, classes, gen):
    c, n = VARIANTS[variant][1]
    coarse = torch.randn(classes, c, n // 4, n // 4, generator=gen)
    return F.interpolate(coarse, size=(n, n), mode="bilinear", align_corners=False)        # smooth class templates


def train_graphed(a, model, templates, classes, crit, gen, dev):
    """The whole step as one hipGraph (cnn_with_pde_amd.graphs + torch's capturable AdamW): the PDE layers' calls are
    launches only once their checkpoint plans are explicit, so nothing in the step needs the host.  The plans are
    re-derived from the current coefficients every --log-every steps and the step is captured again if they changed."""
    static_x = torch.zeros(a.batch, *templates.shape[1:], device=dev)
    static_y = torch.zeros(a.batch, dtype=torch.long, device=dev)
    opt = torch.optim.AdamW(model.parameters(), lr=a.lr, weight_decay=1e-4, capturable=True)
    state = {}

    def one_step():
        opt.zero_grad(set_to_none=False)
        out = model(static_x)
        loss = crit(out, static_y)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
        opt.step()
        return loss, out

    def capture():
        plans = P.freeze_checkpoint_plans(model)           # from the parameters alone: no forward pass, no side effects
        state["plans"] = {id(k): v for k, v in plans.items()}
        state["step"] = P.GraphedStep(one_step)

    def fill():
        labels = torch.randint(0, classes, (a.batch,), generator=gen, device=dev)
        static_y.copy_(labels)
        static_x.copy_(templates[labels] + 1.0 * torch.randn(a.batch, *templates.shape[1:], generator=gen, device=dev))

    fill()
    capture()
    log, recaptures = [], 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for step in range(a.steps):
        fill()
        loss, out = state["step"]()
        if step % a.log_every == 0 or step == a.steps - 1:
            acc = (out.argmax(1) == static_y).float().mean().item()
            log.append({"step": step, "loss": round(loss.item(), 4), "acc": round(acc, 3)})
            print(f"step {step:5d}  loss {loss.item():.4f}  acc {acc:.3f}", flush=True)
            now = {id(k): v for k, v in P.freeze_checkpoint_plans(model).items()}
            if now != state["plans"]:                      # coefficients grew past the frozen plan's margin
                capture()
                recaptures += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"variant": a.variant, "n_gpus": 1, "steps": a.steps, "global_batch": a.batch, "hipgraph": True,
                      "recaptures": recaptures, "samples_per_s": a.batch * a.steps / dt, "first": log[0], "last": log[-1]}))
    return log


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", choices=sorted(VARIANTS), default="mnist")
    ap.add_argument("--amp", action="store_true", help="fp16 autocast + GradScaler (cifar10.py:440,458-467)")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch")
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--log-every", type=int, default=50)
    ap.add_argument("--graph", action="store_true",
                    help="capture the whole training step (forward, loss, backward, clipping, AdamW) in a hipGraph and replay "
                         "it; checkpoint plans frozen and re-checked every --log-every steps.  This is the DEFAULT for a "
                         "single-GPU fp32 run (the reference's own shapes are bound by the host's launch path when run "
                         "eagerly: 1.8-2.4x fewer samples per second)")
    ap.add_argument("--eager", action="store_true", help="one eager autograd step per iteration (always with --amp or several ranks)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU: the PDE layer has no CPU path")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    torch.manual_seed(0)                                  # identical initial weights on every rank
    make, _, classes = VARIANTS[a.variant]
    model = make().to(dev)
    if a.variant == "emotion_model":        # PDELayer's default parameters exceed the explicit stability limit
        with torch.no_grad():
            for n, v in dict(alpha_w1=0.05, alpha_w2=0.02, alpha_w3=-0.01, beta_w1=0.04, beta_w2=0.015, beta_w3=0.01).items():
                getattr(model.pde, n).fill_(v)
    scaler = torch.amp.GradScaler("cuda", enabled=a.amp)
    templates = synthetic_task(a.variant, classes, torch.Generator().manual_seed(7)).to(dev)
    opt = torch.optim.AdamW(model.parameters(), lr=a.lr, weight_decay=1e-4)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=a.steps)
    crit = nn.CrossEntropyLoss(label_smoothing=0.1)
    bucket = P.GradBucket(model.parameters()) if world > 1 else None
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)          # a different shard per rank

    if a.graph and (world > 1 or a.amp):
        raise SystemExit("--graph: single GPU, fp32")
    if a.graph or not (a.eager or world > 1 or a.amp):
        return train_graphed(a, model, templates, classes, crit, gen, dev)

    log, t0 = [], time.perf_counter()
    warm = min(20, a.steps // 4)                          # first steps: library load, allocator and ring warm-up
    for step in range(a.steps):
        if step == warm:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        labels = torch.randint(0, classes, (a.batch,), generator=gen, device=dev)
        x = templates[labels] + 1.0 * torch.randn(a.batch, *templates.shape[1:], generator=gen, device=dev)
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16, enabled=a.amp):
            out = model(x)
            loss = crit(out, labels)
        scaler.scale(loss).backward()
        if bucket is not None:
            bucket.allreduce(average=True)               # ONE collective per step (of the scaled gradients)
        scaler.unscale_(opt)
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
        scaler.step(opt)
        scaler.update()
        sched.step()
        if step % a.log_every == 0 or step == a.steps - 1:
            acc = (out.argmax(1) == labels).float().mean().item()
            log.append({"step": step, "loss": round(loss.item(), 4), "acc": round(acc, 3)})
            if rank == 0:
                print(f"step {step:5d}  loss {loss.item():.4f}  acc {acc:.3f}", flush=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"variant": a.variant, "n_gpus": world, "steps": a.steps, "global_batch": a.batch * world,
                          "samples_per_s": a.batch * world * (a.steps - warm) / dt, "first": log[0], "last": log[-1]}))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return log


if __name__ == "__main__":
    main()
