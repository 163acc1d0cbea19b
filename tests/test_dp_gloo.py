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


def _launch(world, worker, timeout):
    """Start `world` ranks on 127.0.0.1 and return what rank 0 puts on the queue.  The rendezvous port is picked by binding
    port 0 and releasing it, which another process can win in between: one more attempt on a fresh port in that case."""
    import queue
    ctx = mp.get_context("spawn")
    last = None
    for attempt in range(2):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        try:
            got = q.get(timeout=timeout)
        except queue.Empty as e:
            got, last = None, e
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
        if got is not None and all(p.exitcode == 0 for p in procs):
            return got
        last = last or RuntimeError(f"rank exit codes {[p.exitcode for p in procs]}")
    raise last


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
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
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
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
