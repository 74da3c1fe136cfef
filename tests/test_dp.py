"""CPU, world_size 2 over gloo: the bucketed gradient reducer averages gradients exactly as a single process on the
concatenated batch would, including parameters that receive no gradient and multiple buckets."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model():
    torch.manual_seed(0)
    return nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.InstanceNorm2d(8), nn.PReLU(), nn.Conv2d(8, 8, 3, padding=1), nn.Flatten(),
                         nn.Linear(8 * 6 * 6, 5))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from convkan_amd.parallel import BucketedGradReducer
    m = _model()
    unused = nn.Parameter(torch.zeros(3))                       # never gets a gradient
    red = BucketedGradReducer(list(m.parameters()) + [unused], bucket_bytes=2048)     # several small buckets
    assert len(red.buckets) > 2
    g = torch.Generator().manual_seed(7)
    x = torch.randn(4 * world, 3, 6, 6, generator=g)
    t = torch.randint(0, 5, (4 * world,), generator=g)
    for step in range(2):                                       # twice: buckets must re-arm
        m.zero_grad(set_to_none=True)
        nn.functional.cross_entropy(m(x[rank * 4:(rank + 1) * 4]), t[rank * 4:(rank + 1) * 4]).backward()
        red.finish()
    grads = [p.grad.clone() for p in m.parameters()]
    # gradient accumulation: the pass inside no_sync() only accumulates, the armed pass exchanges the sum
    m.zero_grad(set_to_none=True)
    xs, ts = x[rank * 4:(rank + 1) * 4], t[rank * 4:(rank + 1) * 4]
    with red.no_sync():
        nn.functional.cross_entropy(m(xs), ts).backward()
    nn.functional.cross_entropy(m(xs), ts).backward()
    red.finish()
    acc = [p.grad.clone() for p in m.parameters()]
    # a second ARMED pass before finish() must raise instead of reducing a bucket twice
    m.zero_grad(set_to_none=True)
    nn.functional.cross_entropy(m(xs), ts).backward()
    try:
        nn.functional.cross_entropy(m(xs), ts).backward()
        raised = False
    except RuntimeError as e:
        raised = "no_sync" in str(e)
    red.finish()
    # numpy copies travel by value: torch tensors would be shared through file descriptors the parent may open only after this process is gone
    q.put((rank, [g.numpy() for g in grads], unused.grad, [a.numpy() for a in acc], raised))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_bucketed_allreduce_matches_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(world):
        r, grads, ug, acc, raised = q.get(timeout=150)
        got[r] = [torch.from_numpy(g) for g in grads]
        assert ug is None
        assert raised, "a second armed backward pass before finish() did not raise"
        for a, g in zip(acc, got[r]):
            assert torch.allclose(torch.from_numpy(a), 2 * g, rtol=1e-5, atol=1e-7)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    m = _model()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(4 * world, 3, 6, 6, generator=g)
    t = torch.randint(0, 5, (4 * world,), generator=g)
    nn.functional.cross_entropy(m(x), t).backward()
    for r in range(world):
        for a, p in zip(got[r], m.parameters()):
            assert torch.allclose(a, p.grad, rtol=1e-5, atol=1e-7)
