"""FusedAdamW: host layout logic on CPU; the fused step against torch.optim.AdamW -- the optimizer
the reference builds (generic_train.py:24) -- on the GPU."""
import copy

import pytest
import torch
import torch.nn as nn

import convkan_amd as K
from convkan_amd import _lib as L


def _net():
    torch.manual_seed(0)
    return nn.Sequential(nn.Linear(7, 5), nn.PReLU(), nn.Linear(5, 3, bias=False))


def test_flat_layout_and_views():
    net = _net()
    before = [p.detach().clone() for p in net.parameters()]
    opt = K.FusedAdamW(net.parameters(), lr=1e-3, weight_decay=1e-4)
    flat = opt._flat[0]
    assert flat["block"].shape[0] == 3 and flat["n"] % 64 == 0
    base = flat["block"].data_ptr()
    for p, b in zip(net.parameters(), before):
        assert torch.equal(p, b)                                             # values survive the move into the block
        off = p.data_ptr() - base
        assert 0 <= off < flat["n"] * 4 and off % 256 == 0                   # a 256-byte aligned view of row 0
    tab = flat["tab"]                                                        # chunk table of kan_adamw_step_segments
    sizes = [p.numel() for p in net.parameters()]
    assert tab["seg_n"].tolist() == sizes and tab["seg_off"].tolist() == [0, 64, 128, 192]
    assert tab["chunk_seg"].tolist() == [0, 1, 2, 3] and tab["chunk_start"].tolist() == [0, 0, 0, 0]
    big = K.FusedAdamW([nn.Parameter(torch.zeros(20000)), nn.Parameter(torch.zeros(5))])._flat[0]["tab"]
    assert big["chunk_seg"].tolist() == [0, 0, 0, 1] and big["chunk_start"].tolist() == [0, 8192, 16384, 0]
    net(torch.randn(2, 7)).sum().backward()
    assert all(p.grad is not None and p.grad.data_ptr() != 0 for p in net.parameters())
    opt.zero_grad()
    assert all(p.grad is None for p in net.parameters())                     # torch's own zero_grad: nothing to zero per step
    with pytest.raises(L.KanConvError, match="no CPU fallback"):
        net(torch.randn(2, 7)).sum().backward()
        opt.step()


def test_scheduler_and_state_dict_surface():
    net = _net()
    opt = K.FusedAdamW(net.parameters(), lr=1e-3, weight_decay=1e-4)
    sch = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.8)
    sch.step()
    assert abs(opt.param_groups[0]["lr"] - 8e-4) < 1e-12
    sd = opt.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and sd["state"][0]["exp_avg"].shape == (5, 7)
    ref = torch.optim.AdamW(_net().parameters(), lr=1e-3, weight_decay=1e-4)
    assert len(sd["state"]) == len(list(net.parameters())) and set(sd["param_groups"][0]) >= set(ref.state_dict()["param_groups"][0]) - {
        "amsgrad", "maximize", "foreach", "capturable", "differentiable", "fused", "decoupled_weight_decay"}
    sd2 = copy.deepcopy(sd)
    sd2["state"][0]["exp_avg"].fill_(0.5); sd2["state"][0]["step"] = torch.tensor(7.0)
    opt.load_state_dict(sd2)
    assert float(opt._flat[0]["views"][0][1].mean()) == 0.5 and opt.param_groups[0]["step"] == 7
    with pytest.raises(ValueError):
        K.FusedAdamW(_net().parameters(), lr=-1.0)


@pytest.mark.gpu
def test_fused_step_matches_torch_adamw(gpu_lib):
    """Six steps with a decaying learning rate on identical gradients; a parameter without gradient is skipped as torch skips
    it, and bias-corrected with its own step count once its gradients arrive."""
    net = _net()
    ref = copy.deepcopy(net)
    extra_h, extra_r = nn.Parameter(torch.randn(130)), None
    extra_r = nn.Parameter(extra_h.detach().clone())
    dev = net.cuda()
    extra_d = nn.Parameter(extra_h.detach().cuda())
    opt = K.FusedAdamW(list(dev.parameters()) + [extra_d], lr=1e-2, weight_decay=1e-2)
    opt_r = torch.optim.AdamW(list(ref.parameters()) + [extra_r], lr=1e-2, weight_decay=1e-2, foreach=False)
    s1, s2 = (torch.optim.lr_scheduler.ExponentialLR(o, gamma=0.8) for o in (opt, opt_r))
    for it in range(6):
        x = torch.randn(16, 7, generator=torch.Generator().manual_seed(it))
        opt.zero_grad(); opt_r.zero_grad()
        (ref(x) ** 2).mean().backward()
        for pd, pr in zip(dev.parameters(), ref.parameters()):               # the same gradients on both sides: only the update differs
            pd.grad = pr.grad.detach().clone().cuda()
        if it >= 3:                                                          # the extra parameter only gets gradients later
            extra_d.grad = torch.full((130,), 0.1 * it).cuda(); extra_r.grad = torch.full((130,), 0.1 * it)
        else:
            extra_d.grad = None
        opt.step(); opt_r.step(); s1.step(); s2.step()
    for a, b in zip(list(dev.parameters()) + [extra_d], list(ref.parameters()) + [extra_r]):
        assert float((a.detach().cpu() - b.detach()).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-7
    sd, sr = opt.state_dict()["state"], opt_r.state_dict()["state"]
    for i in sr:
        assert float((sd[i]["exp_avg"].cpu() - sr[i]["exp_avg"]).abs().max()) <= 2e-6 * float(sr[i]["exp_avg"].abs().max()) + 1e-9
        assert float((sd[i]["exp_avg_sq"].cpu() - sr[i]["exp_avg_sq"]).abs().max()) <= 2e-6 * float(sr[i]["exp_avg_sq"].abs().max()) + 1e-12


@pytest.mark.gpu
def test_large_block_and_alignment_errors(gpu_lib):
    """A 5 M-element block (grid-stride path, ragged tail) against the closed-form first step; misaligned blocks are refused."""
    import ctypes as C
    n = 5_000_003
    g = torch.Generator(device="cuda").manual_seed(1)
    p, gr = torch.randn(n, device="cuda", generator=g), torch.randn(n, device="cuda", generator=g)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    p0 = p.clone()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ptr = lambda t, o=0: C.c_void_p(t.data_ptr() + 4 * o)
    L.check(L.load().kan_adamw_step(ptr(p), ptr(gr), ptr(m), ptr(v), n, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 1, 1.0, st), "adamw")
    # step 1: m = 0.1 g, v = 0.001 g^2, update = lr * g / (|g| + eps)
    want = p0 * (1 - 1e-5) - 1e-3 * gr / (gr.abs() + 1e-8)
    assert float((p - want).abs().max()) < 1e-6 and float((m - 0.1 * gr).abs().max()) < 1e-6
    assert L.load().kan_adamw_step(ptr(p, 1), ptr(gr), ptr(m), ptr(v), 8, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 1.0, st) != 0
    assert L.load().kan_adamw_step(ptr(p), ptr(gr), ptr(m), ptr(v), 8, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, 1.0, st) != 0


@pytest.mark.gpu
def test_train_model_generic_reduces_loss(gpu_lib):
    """The harness on a small KAN-VGG: AdamW + ExponentialLR as generic_train.py:24-25; the loss on a fixed batch drops."""
    from convkan_amd.models import vggkan
    torch.manual_seed(0)
    m = vggkan(3, 10, arch="VGG11", kan_conv="KAN", classifier_type="Linear", dropout_linear=0.0)
    x, t = torch.randn(8, 3, 32, 32), torch.randint(0, 10, (8,))
    hist = K.train_model_generic(m, [(x, t)] * 3, device="cuda", learning_rate=1e-3, weight_decay=1e-4, gamma=0.8, epochs=4)
    assert len(hist) == 4 and hist[-1] < hist[0] and all(h == h for h in hist)


@pytest.mark.gpu
def test_packed_weights_follow_every_weight_update(gpu_lib, monkeypatch):
    """Single-group layers keep their packed weight layouts between calls (ops._PACKED).  Every way of changing the weights must
    be seen: in-place ops (version counter), FusedAdamW's raw-pointer kernel (it bumps the counters), and writes the counters
    cannot see -- `.data` ops, a raw copy into the storage -- which the device-side fingerprint catches.  The check is a
    fresh, cache-free pack of the same weights (KAN_PACK_CACHE=0 path)."""
    from convkan_amd import ops
    if ops._PACK_CACHE_MAX <= 0:
        pytest.skip("packed-weight cache disabled (KAN_PACK_CACHE=0)")
    torch.manual_seed(0)
    layer = K.KANConv2DLayer(8, 128, 3, padding=1).cuda()
    x = torch.randn(4, 8, 8, 8, device="cuda")

    def uncached():
        keep, ops._PACK_CACHE_MAX = ops._PACK_CACHE_MAX, 0
        try:
            with torch.no_grad():
                return layer(x)
        finally:
            ops._PACK_CACHE_MAX = keep
    ops._PACKED.clear()
    ops.PACK_STATS.update(calls=0, forced=0, skipped=0)
    monkeypatch.setattr(ops, "_PACK_VERIFY_EVERY", 1)                                # fingerprint on every call (cadence: last part of this test)
    with torch.no_grad():
        y0 = layer(x)
        y1 = layer(x)
    assert len(ops._PACKED) == 1 and ops.PACK_STATS == {"calls": 2, "forced": 1, "skipped": 0} and torch.equal(y0, y1) and torch.equal(y0, uncached())
    assert torch.equal(layer(x).detach(), y0)                                        # the training forward shares the layouts (+ wd: one forced pack)
    with torch.no_grad():
        layer.spline_conv[0].weight.mul_(1.5)                                        # version counter moves
        y2 = layer(x)
    assert not torch.equal(y2, y0) and torch.equal(y2, uncached())
    forced = ops.PACK_STATS["forced"]
    layer.spline_conv[0].weight.data.mul_(0.5)                                       # invisible to the version counter
    layer.base_conv[0].weight.data[3, 2, 1, 1] = 7.0                                 # a single element
    with torch.no_grad():
        y3 = layer(x)
    assert ops.PACK_STATS["forced"] == forced                                        # the host saw nothing ...
    assert not torch.equal(y3, y2) and torch.equal(y3, uncached())                   # ... the fingerprint did
    opt = K.FusedAdamW(layer.parameters(), lr=1e-2)
    with torch.no_grad():
        y4 = layer(x)                                                                # parameters now live in the optimizer's flat block
    assert torch.equal(y4, y3)
    layer(x).square().mean().backward()
    opt.step()
    with torch.no_grad():
        y5 = layer(x)
    assert not torch.equal(y5, y4) and torch.equal(y5, uncached())
    # a graph that saved the old bwd-data layout must not be differentiated after the weights moved
    out = layer(x.requires_grad_(True)).square().mean()
    with torch.no_grad():
        layer.spline_conv[0].weight.add_(0.01)
        layer(x)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        out.backward()
    # verification cadence (the default is every 16th call): a stamp-visible write re-packs at once; a raw write is seen at once after
    # ops.weights_changed(), and at the 4th call at the latest with a cadence of 4; unchanged weights cost no device work in between
    monkeypatch.setattr(ops, "_PACK_VERIFY_EVERY", 4)
    with torch.no_grad():
        ya = layer(x)
        calls, skipped = ops.PACK_STATS["calls"], ops.PACK_STATS["skipped"]
        assert torch.equal(layer(x), ya) and torch.equal(layer(x), ya)
        assert ops.PACK_STATS["calls"] == calls and ops.PACK_STATS["skipped"] == skipped + 2
        layer.spline_conv[0].weight.mul_(1.1)                                        # version counter: seen on the very next call
        yb = layer(x)
        assert not torch.equal(yb, ya) and torch.equal(yb, uncached())
        layer.spline_conv[0].weight.data.mul_(0.9)                                   # raw write ...
        ops.weights_changed()                                                        # ... announced
        yc = layer(x)
        assert not torch.equal(yc, yb) and torch.equal(yc, uncached())
        layer.spline_conv[0].weight.data.mul_(1.2)                                   # raw write, not announced: picked up within the cadence
        outs = [layer(x) for _ in range(4)]
        assert torch.equal(outs[-1], uncached()) and not torch.equal(outs[-1], yc)
