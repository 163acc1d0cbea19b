"""Multi-process data parallelism on CPU (gloo, world_size 2 and 4): sharding + one flat gradient
all-reduce reproduces the single-process full-batch gradients.  The compute stand-in is the oracle
(tests may use it); the pieces under test are cnn_with_pde_amd.dist.shard_batch / GradBucket, which
are exactly what bench.py and a DP training loop use on the GPUs."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


PORT_TAKEN = 97          # exit code of a rank whose rendezvous could not bind / connect


def _init(rank, world, port):
    """Rendezvous on 127.0.0.1.  The port was picked by binding port 0 and releasing it, which another process can win in
    between: a rank that cannot rendezvous exits with PORT_TAKEN, the only failure the launcher retries."""
    import sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=60))
    except Exception as e:        # noqa: BLE001  (address in use, connection refused, rendezvous timeout)
        print("rendezvous failed:", repr(e), file=sys.stderr, flush=True)
        os._exit(PORT_TAKEN)


def _launch(world, worker, timeout):
    """Start `world` ranks on 127.0.0.1 and return what rank 0 puts on the queue.  Retried once ONLY when the rendezvous
    port was taken (every failing rank exited with PORT_TAKEN before any work); a rank that crashes or returns non-zero
    for any other reason fails the test at once."""
    import queue
    ctx = mp.get_context("spawn")
    for attempt in range(2):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        try:
            got = q.get(timeout=timeout)
        except queue.Empty:
            got = None
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
                p.join(timeout=10)
        codes = [p.exitcode for p in procs]
        if got is not None and all(c == 0 for c in codes):
            return got
        rendezvous_only = any(c == PORT_TAKEN for c in codes) and all(c in (0, PORT_TAKEN, None, -15) for c in codes)
        if not (rendezvous_only and attempt == 0):
            raise RuntimeError(f"rank exit codes {codes} (attempt {attempt}); result received: {got is not None}")
    raise AssertionError("unreachable")


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    _init(rank, world, port)
    torch.set_num_threads(1)
    from oracle import pde_oracle as O
    import cnn_with_pde_amd as P
    spec = O.cifar10_spec(8, 2, dt=0.05, num_steps=2)
    g = torch.Generator().manual_seed(11)                       # identical on every rank
    params = {k: torch.nn.Parameter(v) for k, v in O.adi_init_params(spec, "cifar10", gen=g).items()}
    B = 6
    u = torch.randn(B, 2, 8, 8, generator=g)
    gy = torch.randn(B, 2, 8, 8, generator=g)
    # loss = sum(y * gy) / B  (a mean over the GLOBAL batch)
    ul, gl = P.shard_batch(u), P.shard_batch(gy)
    y = O.adi_forward(ul, params, spec)
    (y * gl).sum().div(ul.shape[0]).backward()                  # local mean over the shard
    bucket = P.GradBucket(params.values())
    bucket.allreduce(average=True)
    if rank == 0:
        q.put({k: v.grad.clone() for k, v in params.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allreduce_matches_full_batch():
    from oracle import pde_oracle as O
    got = _launch(2, _worker, 120)
    spec = O.cifar10_spec(8, 2, dt=0.05, num_steps=2)
    g = torch.Generator().manual_seed(11)
    params = {k: v.clone().requires_grad_(True) for k, v in O.adi_init_params(spec, "cifar10", gen=g).items()}
    u = torch.randn(6, 2, 8, 8, generator=g)
    gy = torch.randn(6, 2, 8, 8, generator=g)
    (O.adi_forward(u, params, spec) * gy).sum().div(6).backward()
    for k, v in params.items():
        assert torch.allclose(got[k], v.grad, rtol=1e-5, atol=1e-7), k


def _worker4(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    _init(rank, world, port)
    torch.set_num_threads(1)
    from oracle import pde_oracle as O
    import cnn_with_pde_amd as P
    spec = O.cifar10_spec(8, 2, dt=0.05, num_steps=2)
    g = torch.Generator().manual_seed(12)
    params = {k: torch.nn.Parameter(v) for k, v in O.adi_init_params(spec, "cifar10", gen=g).items()}
    B = 7                                                       # ragged: shards of 2, 2, 2, 1
    u = torch.randn(B, 2, 8, 8, generator=g)
    gy = torch.randn(B, 2, 8, 8, generator=g)
    ul, gl = P.shard_batch(u), P.shard_batch(gy)
    assert ul.shape[0] == (2 if rank < 3 else 1)
    # gradients as views of the flat buffer, the collective launched from inside backward by the hooks; per-rank losses
    # are SUMS over the shard (ragged shards), so the ranks' gradients are summed, not averaged
    bucket = P.GradBucket(params.values(), grads_as_views=True)
    bucket.attach_hooks(average=False)
    res = []
    for it in range(2):                                         # twice: the arrival counter and the views survive a step
        bucket.zero()
        y = O.adi_forward(ul, params, spec)
        (y * gl).sum().backward()
        bucket.finish()
        res.append({k: v.grad.clone() for k, v in params.items()})
        assert all(v.grad.data_ptr() == w.data_ptr() for v, w in zip(params.values(), bucket._views()))
    if rank == 0:
        q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_four_ranks_ragged_batch_sum_semantics_views_and_hooks():
    from oracle import pde_oracle as O
    got = _launch(4, _worker4, 180)
    spec = O.cifar10_spec(8, 2, dt=0.05, num_steps=2)
    g = torch.Generator().manual_seed(12)
    params = {k: v.clone().requires_grad_(True) for k, v in O.adi_init_params(spec, "cifar10", gen=g).items()}
    u = torch.randn(7, 2, 8, 8, generator=g)
    gy = torch.randn(7, 2, 8, 8, generator=g)
    (O.adi_forward(u, params, spec) * gy).sum().backward()
    for step in got:
        for k, v in params.items():
            assert torch.allclose(step[k], v.grad, rtol=1e-5, atol=1e-6), k


def _worker_missing(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    _init(rank, world, port)
    torch.set_num_threads(1)
    import cnn_with_pde_amd as P
    a = torch.nn.Parameter(torch.ones(4))
    b = torch.nn.Parameter(torch.ones(3))
    out = {}
    bucket = P.GradBucket([a, b], grads_as_views=True)
    bucket.attach_hooks(average=False)
    # (1) a backward in which one parameter of the bucket gets no gradient: finish() must refuse, not return silently
    bucket.zero()
    (a * (rank + 1)).sum().backward()
    try:
        bucket.finish()
        out["missing"] = "silent"
    except RuntimeError as e:
        out["missing"] = "raised" if "1 of 2" in str(e) else repr(e)
    # (2) the bucket works again afterwards, and the counter starts from zero
    bucket.zero()
    ((a * (rank + 1)).sum() + (b * 2.0).sum()).backward()
    bucket.finish()
    out["after"] = (a.grad.clone(), b.grad.clone())
    # (3) a second backward without finish() in between: the hook refuses (the collective is still pending)
    bucket.zero()
    ((a * 1.0).sum() + (b * 1.0).sum()).backward()
    try:
        ((a * 1.0).sum() + (b * 1.0).sum()).backward()
        out["double"] = "silent"
    except RuntimeError as e:
        out["double"] = "raised" if "finish()" in str(e) else repr(e)
    bucket.finish()
    # (4) and a plain start() twice is refused as well
    bucket.detach_hooks()
    bucket.start(average=False)
    try:
        bucket.start(average=False)
        out["restart"] = "silent"
    except RuntimeError:
        out["restart"] = "raised"
    bucket.finish()
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_hooks_refuse_missing_gradient_and_missed_finish():
    got = _launch(2, _worker_missing, 120)
    assert got["missing"] == "raised", got
    assert got["double"] == "raised", got
    assert got["restart"] == "raised", got
    ga, gb = got["after"]
    assert torch.equal(ga, torch.full((4,), 3.0)) and torch.equal(gb, torch.full((3,), 4.0)), got
