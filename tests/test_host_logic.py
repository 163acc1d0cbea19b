"""Host-side logic of the product package, on CPU: schedules, checkpoint planning, the nn.Module
surface (names, shapes, initial values, state_dict compatibility with the reference's parameters),
and the no-fallback rule."""
import contextlib
import io

import pytest
import torch

import golden_util as G
from oracle import pde_oracle as O
import cnn_with_pde_amd as P


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


@pytest.mark.parametrize("split,mk", [("strang", O.mnist_spec), ("lie", O.cifar2_spec)])
def test_schedule_matches_oracle(split, mk):
    spec = mk(dt=0.013, num_steps=7) if split == "strang" else mk(32, 3, dt=0.013, num_steps=7)
    steps = P.adi_schedule(spec.dt, spec.dx, spec.dy, spec.num_steps, split)
    flat = [(s.axis, s.delta, s.t) for st in steps for s in st]
    assert flat == O.sweep_schedule(spec)           # same Python-double accumulation of current_time


def test_plan_checkpoints():
    assert P.plan_checkpoints([5e-4, 1e-3, 5e-4] * 10) == 0               # cifar/mnist-like: none
    m = P.plan_checkpoints([0.27, 0.54, 0.27] * 4)                        # fashion-like
    assert m != 0 and m < (1 << 11)                                       # never the last sweep (that is y)
    allm = P.plan_checkpoints([3.0] * 6)
    assert allm == 0b11111                                                # huge coefficients: every state kept
    assert P.plan_checkpoints([0.1]) == 0


@pytest.mark.parametrize("name", ["mnist_default", "fashion_default", "svhn_default", "cifar10_default",
                                  "cifar2_default", "tiny_default", "emotion_tame"])
def test_module_surface_matches_reference_parameters(name):
    g = G.Golden(name)
    cls = P.REFERENCE_CLASSES[(g.script, g.cls)]
    layer = quiet(cls, **g.ctor)
    mine = dict(layer.named_parameters())
    assert sorted(mine) == sorted(g.params), (sorted(mine), sorted(g.params))
    for k, v in g.params.items():
        assert tuple(mine[k].shape) == tuple(v.shape), k
    res = layer.load_state_dict({k: v.float() for k, v in g.params.items()}, strict=False)
    assert not res.unexpected_keys
    # initial values of the deterministic parameters equal the reference's (fixtures hold its init)
    fresh = dict(quiet(cls, **g.ctor).named_parameters())
    for k in ("alpha_base", "beta_base", "skip_weight", "channel_scaling", "alpha_w1", "beta_w3"):
        if k in fresh and name.endswith("default"):
            assert torch.equal(fresh[k].detach(), g.params[k].float()), k


def test_reference_attributes_and_helpers():
    m = quiet(P.MnistDiffusionLayer, dx=0.5, dy=2.0)
    for attr in ("size", "dt", "dx", "dy", "num_steps", "stability_eps"):
        assert hasattr(m, attr)
    info = m.get_numerical_stability_info()
    assert set(info) == {"cfl_x", "cfl_y", "dx", "dy", "dt", "stable_x", "stable_y"}
    assert abs(info["cfl_x"] - 2.0 * 0.001 / 0.25) < 1e-9
    a, b = m.get_alpha_beta_at_time(0.3)
    assert a.shape == (28, 28) and float(a.min()) >= m.stability_eps
    e = quiet(P.EnhancedDiffusionLayer, 16, 5)
    a, b = e.get_alpha_beta_at_time(0.0)
    assert float(a.max()) <= 10.0 and e.channel_mixing.shape == (5, 5)
    s = P.SvhnDiffusionLayer(16, 4)
    assert s.channel_coupling.shape == (4, 4) and float(s.skip_weight) == pytest.approx(0.9)
    t = P.ImprovedDiffusionLayer(32, 6)
    assert t.alpha_base.shape == (6,) and t.max_coeff == 0.15 and not hasattr(t, "spatial_modulation")
    p = P.PDELayer(Nx=24, Ny=24, T=0.005)
    assert p.Nt == 5 and p.x.shape == (24,) and "x" in dict(p.named_buffers())


def test_compat_aliases():
    from cnn_with_pde_amd.compat import mnist_test, fashion_mnist, SVHN, cifar10, cifar_2version, tiny_imagenet, \
        emotion_recognition
    assert mnist_test.DiffusionLayer is P.MnistDiffusionLayer
    assert fashion_mnist.DiffusionLayer is P.FashionDiffusionLayer
    assert SVHN.DiffusionLayer is P.SvhnDiffusionLayer
    assert cifar10.EnhancedDiffusionLayer is P.EnhancedDiffusionLayer
    assert cifar_2version.LearnableDiffusionLayer is P.LearnableDiffusionLayer
    assert tiny_imagenet.ImprovedDiffusionLayer is P.ImprovedDiffusionLayer
    assert emotion_recognition.PDELayer is P.PDELayer


def test_no_cpu_fallback():
    """CPU tensors must raise: the product path never routes through a CPU implementation."""
    layer = quiet(P.MnistDiffusionLayer)
    with pytest.raises(P.PdeError):
        layer(torch.zeros(2, 1, 28, 28))
    with pytest.raises(P.PdeError):
        P.channel_mix(torch.zeros(1, 3, 4, 4), torch.eye(3))
    with pytest.raises(P.PdeError):
        P.ImprovedDiffusionLayer(8, 2)(torch.zeros(1, 2, 8, 8))
    with pytest.raises(ValueError):
        layer(torch.zeros(2, 3, 28, 28))            # mnist layer is single-channel


def test_product_does_not_import_oracle():
    import os
    import re
    root = os.path.join(G.GOLDEN_DIR, "..", "..", "cnn-with-pde_amd")
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f


def test_shard_range_covers_batch():
    for total in (1, 7, 512, 4097):
        for world in (1, 2, 3, 8):
            got = [P.shard_range(total, r, world) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
            sizes = [e - b for b, e in got]
            assert max(sizes) - min(sizes) <= 1


def test_counterpart_models_keep_the_reference_names():
    """CPU: every counterpart model constructs, its parameter names are the reference's (the model fixtures made from
    the reference's own modules list them), and the name-selected regulariser sees the parameters it is meant to."""
    import contextlib
    import io
    import golden_util as G
    import cnn_with_pde_amd as P
    for name in G.names(directory=G.MODEL_DIR):
        if name == "model_rh_symmetric_32_train":        # its 3072 x 3072 K is rebuilt from a seed, not stored
            continue
        g = G.Golden(name, G.MODEL_DIR)
        with contextlib.redirect_stdout(io.StringIO()):
            model = P.REFERENCE_CLASSES[(g.script, g.cls)](**g.ctor)
        assert {n for n, _ in model.named_parameters()} == set(g.params), name
        for n, p in model.named_parameters():
            assert tuple(p.shape) == tuple(g.params[n].shape), (name, n)
    with contextlib.redirect_stdout(io.StringIO()):
        models = [P.MnistPDEClassifier(), P.FashionPDEClassifier(), P.CIFAR10PDENoConv(), P.EmotionDiffusionClassifier(),
                  P.HybridPDEExtractor(input_size=8, channels=3)]
    assert "diff.alpha_base" in dict(models[0].named_parameters())
    assert "feature_extractor.pde2.channel_mixing" in dict(models[2].named_parameters())
    hyb = models[4]
    reg = P.hybrid_pde_regularization(hyb, 1e-4, 1e-4, 1e-6)
    want = 0.0
    for n, p in hyb.named_parameters():
        if n.endswith("alpha_base") or n.endswith("beta_base"):
            want += 1e-6 * float(torch.norm(p, p=2) ** 2)
        elif n.endswith("channel_mixing"):
            want += 1e-4 * float(torch.norm(p - torch.eye(3), p="fro") ** 2)
        elif n.endswith("K.weight"):
            want += 1e-4 * float(torch.norm(p, p=2) ** 2)
        elif n.endswith("combination_weights"):
            want += 1e-4 * float(torch.norm(p, p=1))
    assert abs(float(reg) - want) <= 1e-5 * want          # fp32 sums in another order (random initial weights)
