"""Data-parallel row on the GPU box: the bucketed reducer on a REAL RCCL communicator (backend "nccl"), and bench.py's own
rank launcher.  A 1-GPU lease offers world size 1; `always_reduce=True` still sends every bucket through ncclAllReduce(AVG) on
the side stream, with the weight-gradient kernels writing straight into the buckets (ops.GRAD_SINKS)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture()
def rccl_group():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert not dist.is_initialized()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        yield
    finally:
        dist.destroy_process_group()


def test_kan_vgg11_step_through_rccl_reducer(gpu_lib, rccl_group):
    from convkan_amd.models import vggkan
    from convkan_amd.parallel import BucketedGradReducer
    assert dist.get_backend() == "nccl"
    torch.manual_seed(3)
    m = vggkan(3, 10, arch="VGG11", kan_conv="KAN", dropout_linear=0.0).cuda().train()
    x = torch.randn(64, 3, 32, 32, device="cuda")
    t = torch.randint(0, 10, (64,), device="cuda")
    F.cross_entropy(m(x), t).backward()
    plain = {n: p.grad.clone() for n, p in m.named_parameters()}
    m.zero_grad(set_to_none=True)
    red = BucketedGradReducer(m.parameters(), always_reduce=True)
    try:
        assert red.avg_in_collective and red.cuda and len(red.buckets) >= 4      # 332 MB in <= 96 MB buckets
        launched = []
        orig = red._launch
        red._launch = lambda b: (launched.append(b), orig(b))[1]
        for _ in range(2):                                                       # twice: buckets re-arm, sinks are re-offered
            m.zero_grad(set_to_none=True)
            F.cross_entropy(m(x), t).backward()
            assert any(b.work is not None for b in red.buckets)                  # collectives were enqueued during backward
            red.finish()
            torch.cuda.synchronize()
            views = {id(p): v for b in red.buckets for p, v in zip(b.params, b.views)}
            for n, p in m.named_parameters():
                assert p.grad.data_ptr() == views[id(p)].data_ptr(), n
                if p.dim() == 4:                                                 # conv weights: deterministic kernels, AVG over 1 rank
                    assert torch.equal(p.grad, plain[n]), n
                else:                                                            # PReLU slopes (float atomics), head
                    assert float((p.grad - plain[n]).abs().max()) <= 1e-5 * float(plain[n].abs().max()) + 1e-12, n
        assert len(launched) == 2 * len(red.buckets)
    finally:
        red.remove()


def test_shared_layer_does_not_alias_its_sink(gpu_lib):
    """A Parameter feeding two graph nodes of one backward pass: the second node must not write the sink the first one
    filled (autograd would sum two aliases of the last gradient)."""
    import convkan_amd as K
    from convkan_amd.parallel import BucketedGradReducer
    torch.manual_seed(0)
    layer = K.KANConv2DLayer(8, 8, 3, padding=1).cuda()
    x = torch.randn(4, 8, 8, 8, device="cuda")
    layer(layer(x)).square().mean().backward()
    plain = [p.grad.clone() for p in layer.parameters()]
    layer.zero_grad(set_to_none=True)
    red = BucketedGradReducer(layer.parameters())
    try:
        layer(layer(x)).square().mean().backward()
        red.finish()
        for p, g in zip(layer.parameters(), plain):
            assert float((p.grad - g).abs().max()) <= 1e-6 * float(g.abs().max()) + 1e-12
    finally:
        red.remove()


@pytest.mark.timeout(600)
def test_bench_launches_its_own_ranks(gpu_lib):
    """`bench.py --spawn --force-dp`: parent without GPU state -> torch.distributed.run child -> one rank on RCCL; one JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--force-dp", "--steps", "3", "--warmup", "2",
                        "--no-cpu-baseline", "--no-aux"], capture_output=True, text=True, timeout=560, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    doc = json.loads(lines[0])
    assert doc["rccl_ranks"] == 1 and doc["n_gpus"] == 1 and doc["launcher"].startswith("bench.py")
    assert doc["value"] > 1000 and doc["allreduce"]["bytes_per_step"] == 82964690 * 4
