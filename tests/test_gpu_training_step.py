"""The layer inside an end-to-end optimiser step: examples/train_synthetic.py for a
few dozen steps on cuda:0 — the loss must fall and the layer's own parameters must move."""
import importlib.util
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    spec = importlib.util.spec_from_file_location("train_synthetic", os.path.join(ROOT, "examples", "train_synthetic.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("variant,amp", [("mnist", False), ("cifar10", False), ("cifar10_noconv", True), ("svhn_model", False),
                                         ("fashion_model", False)])
def test_training_reduces_loss(variant, amp, monkeypatch):
    mod = _load()
    monkeypatch.setattr(sys, "argv", ["train_synthetic.py", "--variant", variant, "--steps", "60", "--batch", "64",
                                      "--log-every", "59", "--eager"] + (["--amp"] if amp else []))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    log = mod.main()
    assert log[-1]["loss"] < 0.8 * log[0]["loss"], log
    assert log[-1]["acc"] > 0.5, log


@pytest.mark.parametrize("variant", ["mnist", "cifar10_noconv"])
def test_whole_step_replayed_from_a_hipgraph_trains_like_the_eager_loop(variant, monkeypatch):
    """The default of a single-GPU fp32 run (also --graph): forward, loss, backward, clipping and AdamW captured once
    (checkpoint plans frozen from the parameters) and replayed."""
    mod = _load()
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["train_synthetic.py", "--variant", variant, "--steps", "80", "--batch", "64",
                                      "--log-every", "79"] + (["--graph"] if variant == "mnist" else []))
    log = mod.main()
    assert log[-1]["loss"] < 1.2 and log[-1]["acc"] > 0.6, log          # (the first entry is logged after the warm-up steps)


def test_layer_parameters_receive_updates():
    import cnn_with_pde_amd as P
    mod = _load()
    torch.manual_seed(0)
    model = mod.MnistLike().cuda()
    before = {n: p.detach().clone() for n, p in model.diff.named_parameters()}
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2)
    x = torch.randn(32, 1, 28, 28, device="cuda")
    y = torch.randint(0, 10, (32,), device="cuda")
    torch.nn.functional.cross_entropy(model(x), y).backward()
    opt.step()
    moved = {n: float((p.detach() - before[n]).abs().max()) for n, p in model.diff.named_parameters()}
    assert all(v > 0 for v in moved.values()), moved
    assert isinstance(model.diff, P.MnistDiffusionLayer)


def test_grad_bucket_round_trip_on_gpu():
    """GradBucket without a process group: gather -> (no collective) -> scatter leaves every .grad as it was,
    through the one-launch cat / fused foreach copy paths; parameters without a gradient get zeros."""
    import cnn_with_pde_amd as P
    mod = _load()
    torch.manual_seed(1)
    model = mod.MnistLike().cuda()
    x = torch.randn(16, 1, 28, 28, device="cuda")
    model(x).square().mean().backward()
    model.fc2.bias.grad = None
    before = {n: (p.grad.clone() if p.grad is not None else None) for n, p in model.named_parameters()}
    bucket = P.GradBucket(model.parameters())
    bucket.allreduce(average=True)
    for n, p in model.named_parameters():
        if before[n] is None:
            assert p.grad is not None and float(p.grad.abs().max()) == 0.0
        else:
            assert torch.equal(p.grad, before[n]), n
    assert bucket.nbytes() == 4 * sum(p.numel() for p in model.parameters())
