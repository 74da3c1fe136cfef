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


def test_second_step_after_a_weight_update_fills_the_buckets_afresh(gpu_lib, rccl_group):
    """always_reduce=True, a weight changed BETWEEN two steps (an optimizer-style in-place update): after finish() the sinks are
    re-offered, the changed weight is re-packed, and the second step's bucket contents are the gradients of the NEW weights -- not a
    stale bucket, not a stale packed layout.  Also runs one step under trace_step() and checks the diagnostics bench.py reports."""
    import convkan_amd as K
    from convkan_amd.parallel import BucketedGradReducer
    torch.manual_seed(5)
    net = torch.nn.Sequential(K.KANConv2DLayer(4, 128, 3, padding=1, base_activation=torch.nn.SiLU),
                              K.KANConv2DLayer(128, 256, 3, padding=1, base_activation=torch.nn.SiLU)).cuda()
    x = torch.randn(8, 4, 8, 8, device="cuda")

    def plain_grads():
        net.zero_grad(set_to_none=True)
        net(x).square().mean().backward()
        return [p.grad.clone() for p in net.parameters()]
    red = BucketedGradReducer(net.parameters(), bucket_bytes=1 << 20, always_reduce=True)
    try:
        assert len(red.buckets) >= 2
        net.zero_grad(set_to_none=True)
        net(x).square().mean().backward()
        red.finish()
        torch.cuda.synchronize()
        first = [b.flat.clone() for b in red.buckets]
        with torch.no_grad():                                           # the update: every conv weight moves, through the bucket views' owners
            for p in net.parameters():
                if p.dim() == 4:
                    p.add_(0.05 * torch.randn_like(p))
        net.zero_grad(set_to_none=True)
        out = {}
        loss = net(x).square().mean()
        with red.trace_step(out):
            out["t0"] = torch.cuda.Event(enable_timing=True); out["t0"].record(torch.cuda.current_stream())
            loss.backward()
            red.finish()
        second = [b.flat.clone() for b in red.buckets]
        views = {id(p): v for b in red.buckets for p, v in zip(b.params, b.views)}
        got = [p.grad.clone() for p in net.parameters()]
        assert all(p.grad.data_ptr() == views[id(p)].data_ptr() for p in net.parameters())
        assert any(not torch.equal(a, b) for a, b in zip(first, second))
        # diagnostics: every bucket traced once, in launch order, collectives end after they became ready; exposed tail measured
        assert sorted(d["bucket"] for d in out["buckets"]) == list(range(len(red.buckets)))
        assert all(d["end_ms"] >= d["ready_ms"] >= 0.0 for d in out["buckets"]) and out["exposed_ms"] >= 0.0 and out["backward_end_ms"] > 0.0
    finally:
        red.remove()
    want = plain_grads()                                                # the same (new) weights without any reducer
    for (n, p), a, b in zip(net.named_parameters(), got, want):
        if p.dim() == 4:
            assert torch.equal(a, b), n                                 # deterministic kernels, AVG over one rank
        else:
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-12, n


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
    ar = doc["allreduce"]                                                # the N > 1 diagnostics (one untimed traced step, per-rank clocks)
    assert "trace_error" not in ar and ar["exposed_ms"] >= 0.0 and len(ar["bucket_trace"]) == ar["buckets"] >= 4
    assert all(b["end_ms"] >= b["ready_ms"] for b in ar["bucket_trace"]) and doc["ms_per_step_ranks"]["min"] <= doc["ms_per_step_ranks"]["max"]


def test_two_ranks_on_one_gpu_average_conv_kan_gradients(gpu_lib, tmp_path):
    """Two data-parallel ranks of KAN-VGG11 (32 images each) against ONE process on the 64 images: after `finish()` every rank holds the full-batch
    gradient.  A 1-GPU lease has no second device for RCCL, so the two ranks share cuda:0 and exchange over gloo (tests/_dp_rank_gpu.py) -- everything
    above one rank except the RCCL transport itself: hooks of conv-KAN layers arming buckets, gradient sinks under two concurrent processes, the
    side-stream collective, the mean."""
    import torch.nn.functional as F
    from convkan_amd.models import vggkan
    out = str(tmp_path / "grads")
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dp_rank_gpu.py"), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-2000:] for l in logs)
    torch.manual_seed(3)
    m = vggkan(3, 10, arch="VGG11", kan_conv="KAN", dropout_linear=0.0).cuda().train()
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(64, 3, 32, 32, device="cuda", generator=g)
    t = torch.randint(0, 10, (64,), device="cuda", generator=g)
    # reference: the two shards one after the other in THIS process (the launch shapes of the ranks, hence their exact per-shard gradients), then the
    # mean -- a single 64-image launch would differ at model level by the pool / PReLU re-routing of tests/test_gpu_models.py, not by the exchange
    shard = []
    for r in range(2):
        m.zero_grad(set_to_none=True)
        F.cross_entropy(m(x[r * 32:(r + 1) * 32]), t[r * 32:(r + 1) * 32]).backward()
        torch.cuda.synchronize()
        shard.append({n: p.grad.detach().clone() for n, p in m.named_parameters()})
    ranks = [torch.load(f"{out}.rank{r}") for r in range(2)]
    worst = 0.0
    for n, _ in m.named_parameters():
        ref = ((shard[0][n] + shard[1][n]) * 0.5).cpu()
        assert torch.equal(ranks[0][n], ranks[1][n]), n                       # both ranks hold the same reduced gradient
        worst = max(worst, float((ranks[0][n] - ref).abs().max() / (ref.abs().max() + 1e-30)))
    assert worst <= 1e-5, worst                                               # sum over ranks, halved: the mean of the two shard gradients (measured 1.0e-6: the
                                                                              # PReLU-slope and affine-norm gradients use float atomics, everything else is bit-equal)
