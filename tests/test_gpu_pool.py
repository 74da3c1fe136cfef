"""Fused MaxPool2d(k, s) behind the InstanceNorm(+PReLU) kernels (kan_instnorm_prelu_poolk_fwd / _bwd; the AlexNet pattern MaxPool2d(3, 2),
/root/reference models/kan_alexnet.py:120-126) against the same layer followed by torch's max_pool2d: forward values, the routing of the
pooled gradient through OVERLAPPING windows (an element picked by several windows sums their gradients), ties (first maximum in scan order),
trailing rows / columns no window covers."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # kind, C, O, H, W, kernel, padding, (pool k, s), layer kwargs
    ("cheby", 4, 8, 13, 13, 3, 1, (3, 2), dict(degree=4, affine=True)),
    ("cheby", 3, 6, 27, 27, 5, 2, (3, 2), dict(degree=3)),
    ("cheby", 3, 8, 55, 55, 3, 1, (3, 2), dict(degree=4, affine=True)),          # > 1024 pixels per plane
    ("kan", 4, 8, 13, 13, 3, 1, (3, 2), dict()),
    ("kan", 3, 6, 12, 14, 3, 1, (3, 1), dict(affine=True)),
    ("kan", 4, 4, 13, 11, 3, 1, (2, 2), dict()),                                   # 2x2 on an odd plane: the general kernels
    ("kan", 4, 4, 12, 12, 3, 1, (2, 2), dict()),                                   # 2x2 on an even plane: the register kernels
    ("lucas", 4, 6, 17, 17, 3, 1, (5, 3), dict(degree=3)),
    ("kan", 4, 8, 13, 13, 3, 1, (3, 2), dict(groups=2)),
]


def _layer(kind, C, O, k, p, kw):
    import convkan_amd as K
    cls = {"cheby": K.ChebyKANConv2DLayer, "kan": K.KANConv2DLayer, "lucas": K.LucasKANConv2DLayer}[kind]
    return cls(C, O, k, padding=p, **kw)


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}_{c[1]}x{c[2]}_{c[3]}x{c[4]}_pool{c[7][0]}s{c[7][1]}" + ("_g2" if c[8].get("groups") else ""))
def test_fused_pool_matches_layer_then_max_pool2d(case, gpu_lib):
    kind, C, O, H, W, k, p, (pk, ps), kw = case
    torch.manual_seed(3)
    layer = _layer(kind, C, O, k, p, kw).cuda().train()
    x = torch.randn(4, C, H, W, device="cuda")
    x[1] = 0.0                                               # a constant image: every normalised plane is all zeros => every window is a tie
    x[2, :, : H // 2] = x[2, :, H - H // 2:].flip(1)         # mirrored rows: ties between distinct positions of a window
    res = []
    for fused in (True, False):
        layer.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        y = layer(xi, pool=(pk, ps)) if fused else F.max_pool2d(layer(xi), pk, ps)
        g = torch.sin(torch.arange(y.numel(), device="cuda", dtype=torch.float32) * 0.37).view_as(y)
        y.backward(g)
        torch.cuda.synchronize()
        res.append((y.detach(), xi.grad, {n: q.grad.clone() for n, q in layer.named_parameters() if q.grad is not None}))
    (yf, dxf, gf), (yu, dxu, gu) = res
    assert yf.shape == yu.shape == (4, O, (H + 2 * p - k + 1 - pk) // ps + 1, (W + 2 * p - k + 1 - pk) // ps + 1)
    rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-30))
    assert rel(yf, yu) <= 2e-6, rel(yf, yu)
    assert rel(dxf, dxu) <= 2e-5, rel(dxf, dxu)
    for n in gu:
        assert rel(gf[n], gu[n]) <= 2e-5, (n, rel(gf[n], gu[n]))


def test_alexnet_fused_pooling_equals_unfused(gpu_lib):
    """models/kan_alexnet.py: forward_features with the three MaxPool2d(3, 2) fused against the plain nn.Sequential order."""
    from convkan_amd.models import alexnet_kan
    torch.manual_seed(0)
    m = alexnet_kan(num_classes=10, kan_conv="ChebyKAN", degree=4).cuda().eval()
    x = torch.randn(2, 3, 224, 224, device="cuda")
    out = {}
    for fused in (True, False):
        m.fuse_pool = fused
        m.zero_grad(set_to_none=True)
        y = m(x)
        y.square().sum().backward()
        torch.cuda.synchronize()
        out[fused] = (y.detach().clone(), {n: q.grad.clone() for n, q in m.named_parameters()})
    rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-30))
    assert rel(out[True][0], out[False][0]) <= 1e-5
    worst = max(rel(out[True][1][n], out[False][1][n]) for n in out[False][1])
    assert worst <= 1e-3, worst                              # (a near-tie of a window flips under 1-ulp differences of the two norm kernels)


def test_fused_pool_rejects_bad_windows(gpu_lib):
    import convkan_amd as K
    from convkan_amd import _lib as L
    layer = K.KANConv2DLayer(3, 4, 3, padding=1).cuda()
    x = torch.randn(1, 3, 8, 8, device="cuda")
    with pytest.raises(L.KanConvError):
        layer(x, pool=(3, 4))                                # stride > kernel
    with pytest.raises(L.KanConvError):
        layer(x, pool=(9, 2))                                # window larger than the plane


@pytest.mark.parametrize("seed", range(12))
def test_fused_pool_fuzz(seed, gpu_lib):
    """Random (kernel, stride, plane, channel) draws of the fused pool against layer + max_pool2d: window classes ceil(k / s)^2 in {1, 4, 9, 16, 25}, planes with
    trailing rows / columns no window covers, planes above and below the LDS / register-kernel limits."""
    import random
    import convkan_amd as K
    rnd = random.Random(1000 + seed)
    pk = rnd.randint(2, 5)
    ps = rnd.randint(1, pk)
    H, W = rnd.randint(pk + 1, 40), rnd.randint(pk + 1, 40)
    C, O = rnd.choice([1, 3, 4]), rnd.choice([4, 8, 12])
    kind = rnd.choice(["kan", "cheby"])
    B = rnd.choice([1, 3, 5])
    torch.manual_seed(seed)
    layer = (K.KANConv2DLayer(C, O, 3, padding=1, base_activation=torch.nn.SiLU) if kind == "kan" else K.ChebyKANConv2DLayer(C, O, 3, padding=1, degree=3, affine=True)).cuda().train()
    x = torch.randn(B, C, H, W, device="cuda")
    res = []
    for fused in (True, False):
        layer.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        y = layer(xi, pool=(pk, ps)) if fused else F.max_pool2d(layer(xi), pk, ps)
        g = torch.cos(torch.arange(y.numel(), device="cuda", dtype=torch.float32) * 0.73).view_as(y)
        y.backward(g)
        torch.cuda.synchronize()
        res.append((y.detach(), xi.grad, {n: q.grad.clone() for n, q in layer.named_parameters() if q.grad is not None}))
    (yf, dxf, gf), (yu, dxu, gu) = res
    rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-30))
    tag = f"{kind} C{C} O{O} {H}x{W} B{B} pool {pk}/{ps}"
    assert yf.shape == yu.shape, tag
    assert rel(yf, yu) <= 2e-6 and rel(dxf, dxu) <= 2e-5, (tag, rel(yf, yu), rel(dxf, dxu))
    for n in gu:
        assert rel(gf[n], gu[n]) <= 2e-5, (tag, n, rel(gf[n], gu[n]))
