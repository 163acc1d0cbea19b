"""Per-GPU timing of the other BASELINE.json configurations with the
algorithmic-bytes roofline fraction, for DESIGN.md.  Not the headline bench (that is bench.py)."""
import contextlib, io, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cnn_with_pde_amd as P


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def run(name, layer, shape, dtype, bytes_per_elem, steps=20, warm=3):
    layer = layer.cuda()
    g = torch.Generator().manual_seed(1)
    u = torch.randn(*shape, generator=g).to(dtype).cuda().requires_grad_(True)
    gy = torch.randn(*shape, generator=g).to(dtype).cuda()
    def step():
        for p in layer.parameters():
            p.grad = None
        u.grad = None
        layer(u).backward(gy)
    for _ in range(warm):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    elems = u.numel()
    out = {"config": name, "shape": list(shape), "dtype": str(dtype).replace("torch.", ""), "ms_per_step": dt * 1e3,
           "Msamples_per_s": shape[0] / dt / 1e6, "algorithmic_GBps": elems * bytes_per_elem / dt / 1e9,
           "hbm_frac_of_8TBps": elems * bytes_per_elem / dt / 8e12}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["cfg3", "cfg3c1", "cfg4", "cfg5", "cfg1", "emotion"]
    if "cfg1" in which:      # mnist plumbing config (one CU-full of planes; latency-bound)
        run("cfg1 mnist (64,1,28,28)", quiet(P.MnistDiffusionLayer), (64, 1, 28, 28), torch.float32, 20)
    if "cfg3" in which:      # fashion semantics broadcast to 32 channels = SVHN layer with coupling I, skip -40
        L = P.SvhnDiffusionLayer(28, 32, dt=0.3, num_steps=4)
        with torch.no_grad():
            L.alpha_base.fill_(1.8); L.beta_base.fill_(1.8); L.alpha_time_coeff.zero_(); L.beta_time_coeff.zero_()
            L.channel_coupling.copy_(torch.eye(32)); L.skip_weight.fill_(-40.0)
        run("cfg3 fashion semantics x32ch via SVHN layer (512,32,28,28), per-step coupling launches", L,
            (512, 32, 28, 28), torch.float32, 20)
    if "cfg3c1" in which:
        run("cfg3 literal fashion_mnist.DiffusionLayer (4096,1,28,28)", P.FashionDiffusionLayer(), (4096, 1, 28, 28),
            torch.float32, 20)
    if "cfg4" in which:      # SVHN semantics, 128 channels, 20 steps, bf16 I/O
        L = P.SvhnDiffusionLayer(32, 128, num_steps=20)
        with torch.no_grad():
            L.channel_coupling.copy_(torch.eye(128) + 0.01 * torch.randn(128, 128))
        run("cfg4 SVHN.DiffusionLayer(32,128,num_steps=20) bf16 I/O (512,128,32,32)", L, (512, 128, 32, 32),
            torch.bfloat16, 10, steps=5, warm=2)
    if "cfg5" in which:
        run("cfg5 tiny_imagenet.ImprovedDiffusionLayer(64,64) (256,64,64,64)", P.ImprovedDiffusionLayer(64, 64),
            (256, 64, 64, 64), torch.float32, 20)
    if "emotion" in which:
        run("emotion_recognition.PDELayer() (64,1,48,48)", P.PDELayer(), (64, 1, 48, 48), torch.float32, 20)
