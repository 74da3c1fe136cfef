"""GPU parity of the HIP path against the CPU oracle on seeded inputs (shapes of BASELINE.json's configs at reduced
batch, plus edge cases), and size-independent properties checked at the configs' FULL sizes."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import convkan_amd as K
from helpers import check_vs_oracle, relerr

pytestmark = pytest.mark.gpu

VGG11 = [(3, 64, 32), (64, 128, 16), (128, 256, 8), (256, 256, 8), (256, 512, 4), (512, 512, 4), (512, 512, 2), (512, 512, 2)]
ALEX = [(3, 64, 224, 11, 4, 2), (64, 192, 27, 5, 1, 2), (192, 384, 13, 3, 1, 1), (384, 256, 13, 3, 1, 1), (256, 256, 13, 3, 1, 1)]


def _compare(layer, cfg, x_cpu):
    """fwd+bwd on the HIP layer against the fp64 oracle with the same parameters; per-tensor tolerance
    max(stated, 4 x the fp32 oracle's own distance from fp64 on that tensor) -- helpers.check_vs_oracle.  No per-shape multipliers."""
    check_vs_oracle(layer, cfg, x_cpu, groups=cfg.get("groups", 1))


def _cfg(kind, C, O, k=3, s=1, p=1, d=1, groups=1, **kw):
    c = dict(kind=kind, C=C, O=O, k=k, s=s, p=p, d=d, groups=groups)
    c.update(kw)
    return c


@pytest.mark.parametrize("li", range(8))
def test_vgg11_layer_shapes_vs_oracle(li, gpu_lib):
    C, O, H = VGG11[li]
    torch.manual_seed(li)
    layer = K.KANConv2DLayer(C, O, 3, padding=1, base_activation=nn.SiLU)
    B = 8 if H > 2 else 16
    # InstanceNorm over 2x2 planes amplifies rounding noise (the reference's own fp32-vs-fp64 noise there is ~1e-5)
    _compare(layer, _cfg("bspline", C, O, act="silu"), torch.randn(B, C, H, H))


def test_config2_fastkan_full_batch_vs_oracle(gpu_lib):
    torch.manual_seed(2)
    layer = K.FastKANConv2DLayer(3, 64, 3)                       # BASELINE.json configs[1]: 256x3x32x32, ctor-default padding 0
    _compare(layer, _cfg("rbf", 3, 64, p=0), torch.randn(256, 3, 32, 32))


@pytest.mark.parametrize("li", range(5))
def test_config5_cheby_alexnet_layer_shapes_vs_oracle(li, gpu_lib):
    C, O, H, k, s, p = ALEX[li]
    torch.manual_seed(li)
    layer = K.ChebyKANConv2DLayer(C, O, k, degree=4, stride=s, padding=p, affine=True)
    _compare(layer, _cfg("cheby", C, O, k=k, s=s, p=p, degree=4), torch.randn(2, C, H, H))


@pytest.mark.parametrize("case", [
    dict(C=1, O=1, H=1, W=1, B=1), dict(C=2, O=3, H=1, W=9, B=3), dict(C=5, O=70, H=6, W=7, B=2), dict(C=3, O=130, H=5, W=5, B=2),
    dict(C=3, O=192, H=9, W=4, B=2), dict(C=20, O=8, H=12, W=12, B=5, k=5, p=1, s=3), dict(C=4, O=4, H=8, W=8, B=2, groups=4),
    dict(C=6, O=9, H=8, W=8, B=2, groups=3, d=2, p=2), dict(C=3, O=4, H=5, W=5, B=1, k=(3, 1), p=(1, 0))])
def test_edge_shapes_vs_oracle(case, gpu_lib):
    torch.manual_seed(5)
    k, p, s, d, G = case.get("k", 3), case.get("p", 1), case.get("s", 1), case.get("d", 1), case.get("groups", 1)
    layer = K.KANConv2DLayer(case["C"], case["O"], k, padding=p, stride=s, dilation=d, groups=G)
    one_px = case["H"] * case["W"] == 1                          # InstanceNorm of a single value: variance 0, y = 0 on both sides
    x = torch.randn(case["B"], case["C"], case["H"], case["W"]) * 1.5
    if one_px:
        y = layer.cuda()(x.cuda())
        assert float(y.abs().max()) == 0.0
        return
    _compare(layer, _cfg("bspline", case["C"], case["O"], k=k, s=s, p=p, d=d, groups=G), x)


GROUPED = [
    # (kind, C, O, groups, H, B, extra layer kwargs) -- every group runs in the same launches (KanGeom.groups)
    ("bspline", 64, 160, 2, 12, 8, {}),                     # several pixel tiles, ragged 80-output groups
    ("bspline", 40, 40, 40, 14, 16, dict(affine=True)),     # depthwise (kan_mobilenetv2.py:253-255 replace_depthwise), per-group gamma/beta
    ("bspline", 32, 32, 4, 2, 32, {}),                      # position-major 2x2 planes: tap skipping with group offsets
    ("bspline", 32, 48, 2, 4, 32, {}),                      # position-major weight gradient on 4x4 planes
    ("bspline", 24, 24, 24, 4, 64, {}),                     # depthwise on 4x4 planes
    ("rbf", 48, 96, 3, 9, 4, {}),
    ("cheby", 30, 60, 5, 7, 6, dict(affine=True)),
]


@pytest.mark.parametrize("case", GROUPED, ids=lambda c: f"{c[0]}-C{c[1]}-O{c[2]}-g{c[3]}-{c[4]}x{c[4]}")
def test_grouped_single_launch_vs_oracle(case, gpu_lib):
    kind, C, O, G, H, B, kw = case
    torch.manual_seed(C + G)
    if kind == "bspline":
        layer = K.KANConv2DLayer(C, O, 3, groups=G, padding=1, base_activation=nn.SiLU, **kw)
        cfg = _cfg("bspline", C, O, groups=G, act="silu")
    elif kind == "rbf":
        layer = K.FastKANConv2DLayer(C, O, 3, groups=G, padding=1, **kw)
        cfg = _cfg("rbf", C, O, groups=G)
    else:
        layer = K.ChebyKANConv2DLayer(C, O, 3, groups=G, padding=1, degree=3, **kw)
        cfg = _cfg("cheby", C, O, groups=G, degree=3)
    with torch.no_grad():                                   # distinct per-group norm / PReLU parameters
        for n, p in layer.named_parameters():
            if "prelus" in n:
                p.fill_(0.05 + 0.03 * int(n.split(".")[1]))
            elif "layer_norm" in n:
                p.add_(0.2 * torch.randn_like(p))
    # depthwise: each group's PReLU-slope gradient is ONE scalar summed over a single channel with heavy cancellation, so
    # its max-normalised error is the absolute error over that one (possibly small) value
    _compare(layer, cfg, torch.randn(B, C, H, H))


@pytest.mark.parametrize("gs,C,O,H,B", [(8, 64, 128, 8, 16), (7, 32, 256, 6, 8), (6, 40, 70, 9, 4)])
def test_many_planes_single_item_steps(gs, C, O, H, B, gpu_lib):
    """P >= 10 planes leave room for ONE (tap, channel) item per LDS step, so half the forward kernel's waves have nothing to
    stage; they must still wait for their own async weight copy before the step barrier (a wave that skipped the wait
    raced the readers of its rows)."""
    torch.manual_seed(gs)
    layer = K.KANConv2DLayer(C, O, 3, padding=1, grid_size=gs, base_activation=nn.SiLU)
    _compare(layer, _cfg("bspline", C, O, act="silu", grid_size=gs), torch.randn(B, C, H, H))


HALO = [
    # (factory key, C, O, H, B): 3x3 / stride 1 / pad 1 layers on the plane sizes the halo forward kernel serves, one per
    # compile-time spec and tile width (B-spline SiLU / GELU, ChebyKAN degree 3, recurrence degree 3; 128- and 256-output tiles)
    ("KAN", 64, 128, 16, 8, dict(base_activation=nn.SiLU)), ("KAN", 32, 256, 8, 6, dict(base_activation=nn.GELU)),
    ("KAN", 16, 256, 4, 40, dict(base_activation=nn.SiLU)), ("KAN", 6, 128, 32, 3, dict(base_activation=nn.SiLU)),
    ("ChebyKAN", 32, 128, 8, 8, dict(degree=3)), ("LucasKAN", 64, 128, 16, 4, dict(base_activation=nn.SiLU)),
    ("JacobiKAN", 32, 256, 8, 8, dict(base_activation=nn.SiLU)), ("LaguerreKAN", 20, 128, 4, 24, {}),
    # ragged last tile: images missing from the last 128-pixel tile (8x8: 2 images per tile, 4x4: 8 per tile), one image only
    ("KAN", 8, 128, 8, 5, dict(base_activation=nn.SiLU)), ("KAN", 8, 128, 4, 13, dict(base_activation=nn.SiLU)),
    ("KAN", 4, 128, 16, 1, dict(base_activation=nn.SiLU)), ("KAN", 2, 128, 32, 1, dict(base_activation=nn.GELU)),
    # groups folded into the halo launch (per-group x / weight / z offsets)
    ("KAN", 32, 256, 8, 4, dict(base_activation=nn.SiLU, groups=2)), ("LucasKAN", 24, 384, 4, 16, dict(base_activation=nn.SiLU, groups=3)),
    # halo-shaped but NOT halo-served: LegendreKAN evaluates its basis on a second (batch-normalised) tensor, which the
    # one-input halo kernel cannot take -- the plan must keep it on the tap-major kernel and its weight order
    ("LegendreKAN", 16, 128, 8, 4, {}), ("LegendreKAN", 4, 256, 16, 2, {}),
]


@pytest.mark.parametrize("case", HALO, ids=lambda c: f"{c[0]}-C{c[1]}-O{c[2]}-{c[3]}x{c[3]}")
def test_halo_forward_shapes_vs_oracle(case, gpu_lib):
    name, C, O, H, B, kw = case
    torch.manual_seed(C + O + H)
    layer = K.CONV_KAN_FACTORY[name](C, O, 3, **kw)
    act = {nn.SiLU: "silu", nn.GELU: "gelu"}.get(kw.get("base_activation", nn.GELU), "gelu")
    kind = {"KAN": "bspline", "ChebyKAN": "cheby", "LucasKAN": "lucas", "JacobiKAN": "jacobi", "LaguerreKAN": "laguerre",
            "LegendreKAN": "legendre"}[name]
    extra = {"JacobiKAN": {"a": 1.0, "b": 1.0}, "LaguerreKAN": {"alpha": 1.0}}.get(name, {})
    cfg = _cfg(kind, C, O, act=act, degree=3, extra=extra, groups=kw.get("groups", 1))
    if kind in ("laguerre", "legendre"):
        cfg["act"] = "gelu"
    _compare(layer, cfg, torch.randn(B, C, H, H))


def test_nan_and_out_of_grid_inputs(gpu_lib):
    """x outside the knot span has all bases zero (kan_layers.py:209); NaN inputs propagate through the base branch only."""
    torch.manual_seed(1)
    layer = K.KANConv2DLayer(3, 4, 3, padding=1, base_activation=None)
    x = torch.randn(2, 3, 6, 6) * 4.0
    _compare(layer, _cfg("bspline", 3, 4, act="none"), x)
    spec = layer.cuda().conv_spec()
    xn = x.clone(); xn[0, 1, 2, 2] = float("nan")
    z = K.ops.kan_conv(spec, xn.cuda(), None, [layer.base_conv[0].weight], [layer.spline_conv[0].weight])
    assert torch.isnan(z[0]).any() and not torch.isnan(z[1]).any()


# ------------------------------------------------------------------------------------- properties at FULL size
def _stage(layer, x):
    return K.ops.kan_conv(layer.conv_spec(), x, None, [m.weight for m in layer.base_conv], [m.weight for m in layer.spline_conv])


@pytest.mark.parametrize("li", [1, 3, 5, 7])
def test_full_size_properties(li, gpu_lib):
    """BASELINE.json configs[2] layer shapes at bs=256: (a) the conv stage is linear in the weights, (b) Euler identity
    <dz, z> = <dW, W> ties bwd-weight to forward, (c) a sub-batch reproduces the full batch (batch independence),
    (d) <dz, J dx-direction> consistency of bwd-data through a directional derivative of the base branch."""
    C, O, H = VGG11[li]
    torch.manual_seed(li)
    a = K.KANConv2DLayer(C, O, 3, padding=1, base_activation=nn.SiLU).cuda()
    b = K.KANConv2DLayer(C, O, 3, padding=1, base_activation=nn.SiLU).cuda()
    x = torch.randn(256, C, H, H, device="cuda")
    za, zb = _stage(a, x), _stage(b, x)
    s = K.KANConv2DLayer(C, O, 3, padding=1, base_activation=nn.SiLU).cuda()
    with torch.no_grad():
        s.base_conv[0].weight.copy_(a.base_conv[0].weight + b.base_conv[0].weight)
        s.spline_conv[0].weight.copy_(a.spline_conv[0].weight + b.spline_conv[0].weight)
    assert relerr(_stage(s, x), za + zb) <= 1e-5                                     # (a)
    for p in a.parameters():
        p.requires_grad_(True)
    xg = x.clone().requires_grad_(True)
    z = _stage(a, xg)
    dz = torch.randn_like(z)
    z.backward(dz)
    lhs = float((dz.double() * z.detach().double()).sum())
    rhs = float(sum((m.weight.grad.double() * m.weight.detach().double()).sum() for m in list(a.base_conv) + list(a.spline_conv)))
    assert abs(lhs - rhs) <= 2e-5 * max(abs(lhs), float(dz.double().norm() * z.detach().double().norm()) * 1e-2)   # (b)
    zsub = _stage(a, x[40:56].contiguous())
    assert relerr(zsub, z.detach()[40:56]) <= 1e-5                                   # (c)
    # (d) finite-difference check of <dz, z(x + e v) - z(x - e v)> / 2e  against  <dx, v>  (fp64 accumulation of fp32 results)
    v = torch.randn_like(x)
    eps = 1e-2
    fd = float((dz.double() * (_stage(a, x + eps * v).double() - _stage(a, x - eps * v).double())).sum()) / (2 * eps)
    an = float((xg.grad.double() * v.double()).sum())
    assert abs(fd - an) <= 2e-3 * max(abs(an), 1.0), (fd, an)


def test_full_model_step_is_batch_independent(gpu_lib):
    """KAN-VGG11 at bs=256 (configs[2]): per-sample logits do not depend on the rest of the batch (InstanceNorm only)."""
    from convkan_amd.models import vggkan
    torch.manual_seed(0)
    m = vggkan(3, 10, arch="VGG11", kan_conv="KAN", classifier_type="Linear").cuda().eval()
    x = torch.randn(256, 3, 32, 32, device="cuda")
    with torch.no_grad():
        full = m(x)
        part = m(x[100:108].contiguous())
    assert relerr(part, full[100:108]) <= 2e-3
    loss = F.cross_entropy(m(x), torch.randint(0, 10, (256,), device="cuda"))
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


@pytest.mark.parametrize("B", [1, 3, 32, 100])
def test_vgg11_runs_and_is_batch_consistent_at_odd_batch_sizes(B, gpu_lib):
    """Planner edge cases (ragged pixel tiles, split selection, position-major paths with B < 128 or not a power of
    two): KAN-VGG11 logits of a batch equal those of its samples run one at a time."""
    from convkan_amd.models import vggkan
    torch.manual_seed(3)
    m = vggkan(3, 10, arch="VGG11", kan_conv="KAN", classifier_type="Linear").cuda().eval()
    x = torch.randn(B, 3, 32, 32, device="cuda", requires_grad=True)
    y = m(x)
    y.square().sum().backward()
    assert torch.isfinite(y).all() and torch.isfinite(x.grad).all()
    with torch.no_grad():
        single = torch.cat([m(x[i:i + 1].detach()) for i in range(min(B, 3))])
    assert relerr(y[:single.shape[0]], single) <= 2e-3


# ------------------------------------------------------------------------------------- fused MaxPool2d(2, 2)
@pytest.mark.parametrize("case", [(3, 64, 32, 4, {}), (8, 128, 8, 6, {}), (16, 64, 4, 20, {}), (8, 12, 6, 3, dict(groups=2, affine=True)),
                                  (4, 8, 2, 33, {}), (6, 16, 16, 2, dict(base_activation=None))],
                         ids=lambda c: f"C{c[0]}-O{c[1]}-{c[2]}x{c[2]}")
def test_fused_maxpool_equals_layer_then_maxpool(case, gpu_lib):
    """layer(x, pool=True) -- the pooling inside the InstanceNorm+PReLU kernels -- against the same layer followed by torch's
    MaxPool2d(2, 2) (kan_vgg.py:97-101): identical forward values, gradients equal to rounding."""
    C, O, H, B, kw = case
    torch.manual_seed(C + O + H)
    layer = K.KANConv2DLayer(C, O, 3, padding=1, **kw).cuda()
    with torch.no_grad():
        layer.prelus[0].weight.fill_(0.3)
    x = torch.randn(B, C, H, H, device="cuda")
    go = torch.randn(B, O, H // 2, H // 2, device="cuda")
    outs = []
    for fused in (True, False):
        layer.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        y = layer(xi, pool=True) if fused else F.max_pool2d(layer(xi), 2, 2)
        y.backward(go)
        outs.append((y.detach(), xi.grad, {n: p.grad.clone() for n, p in layer.named_parameters()}))
    (y1, dx1, g1), (y0, dx0, g0) = outs
    assert torch.equal(y1, y0)
    assert relerr(dx1, dx0) <= 2e-6
    for n in g0:
        assert relerr(g1[n], g0[n]) <= 2e-6, n
    odd = K.KANConv2DLayer(3, 8, 3, padding=1).cuda()                           # odd planes fall back to the unfused sequence
    xo = torch.randn(2, 3, 7, 7, device="cuda")
    assert torch.equal(odd(xo, pool=True), F.max_pool2d(odd(xo), 2, 2))


def test_vgg_fused_pool_matches_unfused_model(gpu_lib):
    from convkan_amd.models import vggkan
    torch.manual_seed(5)
    m = vggkan(3, 10, arch="VGG11", kan_conv="KAN", classifier_type="Linear", dropout_linear=0.0).cuda().train()
    x, t = torch.randn(6, 3, 32, 32, device="cuda"), torch.randint(0, 10, (6,), device="cuda")
    res = []
    for fuse in (True, False):
        m.fuse_pool = fuse
        m.zero_grad(set_to_none=True)
        logits = m(x)
        F.cross_entropy(logits, t).backward()
        res.append((logits.detach(), [p.grad.clone() for p in m.parameters()]))
    assert relerr(res[0][0], res[1][0]) <= 1e-6
    for a, b in zip(res[0][1], res[1][1]):
        assert relerr(a, b) <= 1e-5


def test_l1_wrapped_layer_runs_on_the_hip_path(gpu_lib):
    """kan_conv.py:66-68 wraps the layer in L1 when l1_decay > 0; the wrapper's full backward hook must coexist with the
    custom autograd functions.  Every gradient is the plain one, or the plain one plus l1_decay * sign(p) where the hook
    found the gradient still empty (utils/regularization.py:79-83)."""
    torch.manual_seed(0)
    w = K.CONV_KAN_FACTORY["KAN"](3, 8, 3, l1_decay=0.1).cuda()
    x = torch.randn(2, 3, 8, 8, device="cuda", requires_grad=True)
    go = torch.randn(2, 8, 8, 8, device="cuda")
    w.module(x).backward(go)
    plain = {n: p.grad.clone() for n, p in w.module.named_parameters()}
    w.module.zero_grad(set_to_none=True)
    y = w(x)
    assert torch.equal(y, w.module(x))
    y.backward(go)
    for n, p in w.module.named_parameters():
        pen = 0.1 * torch.sign(p.detach())
        assert relerr(p.grad, plain[n]) <= 1e-6 or relerr(p.grad, plain[n] + pen) <= 1e-6, n


@pytest.mark.parametrize("fam", ["LucasKAN", "LaguerreKAN", "TaylorKAN"])
def test_fused_maxpool_recurrence_families(fam, gpu_lib):
    """The same fusion for the recurrence-family layers (their VGG variants hit it through VGGKAN.forward_features)."""
    torch.manual_seed(1)
    layer = K.CONV_KAN_FACTORY[fam](6, 128, 3).cuda()
    x = torch.randn(5, 6, 8, 8, device="cuda")
    go = torch.randn(5, 128, 4, 4, device="cuda")
    outs = []
    for fused in (True, False):
        layer.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        y = layer(xi, pool=True) if fused else F.max_pool2d(layer(xi), 2, 2)
        y.backward(go)
        outs.append((y.detach(), xi.grad, [p.grad.clone() for p in layer.parameters()]))
    assert torch.equal(outs[0][0], outs[1][0]) and relerr(outs[0][1], outs[1][1]) <= 2e-6
    for a, b in zip(outs[0][2], outs[1][2]):
        assert relerr(a, b) <= 2e-6


@pytest.mark.parametrize("C,O,H,B,G", [(128, 128, 2, 128, 1), (128, 256, 4, 32, 1), (256, 128, 2, 256, 1), (128, 128, 4, 128, 1), (256, 256, 2, 128, 2),
                                       (256, 256, 4, 16, 2)])
def test_expanded_position_major_kernels_vs_oracle(C, O, H, B, G, gpu_lib):
    """Small padded planes on the expanded position-major operand (kan_position_major_expanded): DMA-only weight gradient
    (C * 9 % 128 == 0, B % 16 == 0) on 4x4 / 2x2 planes, DMA-only forward (B % 128 == 0) on 2x2 -- the paths the KAN-VGG11 layers 4-7
    take at bs 256 -- against the oracle, plus the plan flags that say which kernels ran."""
    from convkan_amd import ops
    torch.manual_seed(C + O + H)
    layer = K.KANConv2DLayer(C, O, 3, padding=1, groups=G, base_activation=nn.SiLU)
    geom, basis, plan = ops._plan_cached(layer.conv_spec(), B, C // G, H, H, O // G, C, O)
    assert plan.bwd_weight_expanded == 1 and plan.e_pm_wanted == 1
    assert plan.fwd_expanded == (1 if (H == 2 and B % 128 == 0) else 0)
    _compare(layer, _cfg("bspline", C, O, groups=G, act="silu"), torch.randn(B, C, H, H))


@pytest.mark.parametrize("C,O,B,act,want", [(6, 256, 8, "silu", 3), (10, 256, 24, "gelu", 3), (4, 512, 16, "silu", 3), (6, 48, 8, "silu", 2),
                                              (6, 256, 12, "silu", 1)],
                         ids=["o256_b8", "o256_b24_gelu", "o512_b16", "o48_b8_bwd_only", "o256_b12_plain_bwd"])
def test_row_block_kernels_on_4x4_planes_vs_oracle(C, O, B, act, want, gpu_lib):
    """4x4 planes: the halo forward (256-output tiles) and bwd-data (batch a multiple of 8, O a multiple of 16) order a tile's pixels
    (row, image, column) and skip the MFMA blocks of rows whose tap row lies outside the plane; bwd-data pairs tap rows 0 and 2 in
    mixed depth steps.  The last two cases take only one of the two kernels (O = 48: 64-output forward tiles; B = 12: plain bwd-data)."""
    torch.manual_seed(O + B)
    from convkan_amd import ops
    layer = K.KANConv2DLayer(C, O, 3, padding=1, base_activation=nn.SiLU if act == "silu" else nn.GELU)
    assert ops._plan_cached(layer.conv_spec(), B, C, 4, 4, O, C, O)[2].row_blocks == want      # bit 0: forward, bit 1: bwd-data
    _compare(layer, _cfg("bspline", C, O, act=act), torch.randn(B, C, 4, 4) * 1.5)


def test_row_block_forward_for_the_recurrence_spec_vs_oracle(gpu_lib):
    """The degree-3 recurrence spec (P = 5 planes) on the same row-ordered 4x4 forward (256-output halo tiles); its bwd-data stays plain."""
    from convkan_amd import ops
    torch.manual_seed(11)
    layer = K.LucasKANConv2DLayer(6, 256, 3, degree=3, padding=1, base_activation=nn.SiLU)
    plan = ops._plan_cached(layer.conv_spec(), 8, 6, 4, 4, 256, 6, 256)[2]
    assert plan.fwd_halo == 1 and plan.row_blocks == 1
    _compare(layer, _cfg("lucas", 6, 256, act="silu", degree=3), torch.randn(8, 6, 4, 4))


@pytest.mark.parametrize("C,O,H,B,G,act", [(6, 32, 8, 4, 1, "silu"), (8, 128, 4, 16, 1, "gelu"), (4, 256, 16, 2, 1, "silu"), (8, 64, 2, 32, 2, "silu"), (3, 8, 7, 3, 1, "none")],
                         ids=["8x8", "4x4_o128", "16x16_o256", "2x2_groups2", "identity_7x7"])
def test_constant_plane_spec_vs_oracle(C, O, H, B, G, act, gpu_lib):
    """A recurrence family at degree 0 = base branch + ONE constant plane (the compile-time two-plane spec that also carries Wav-KAN's plain
    convolutions); `act='none'` leaves the generic kernels (no base branch)."""
    torch.manual_seed(C * O + H)
    acts = {"silu": nn.SiLU, "gelu": nn.GELU, "none": None}
    layer = K.BesselKANConv2DLayer(C, O, 3, degree=0, padding=1, groups=G, base_activation=acts[act])
    _compare(layer, _cfg("bessel", C, O, groups=G, act=act, degree=0), torch.randn(B, C, H, H))


@pytest.mark.parametrize("fam,G,H", [("relu", 1, 8), ("relu", 2, 4), ("gram", 1, 8), ("gram", 2, 16)], ids=["relu", "relu_g2_4x4", "gram", "gram_g2_16x16"])
def test_parameter_gradients_from_the_input_gradient_launch_match_the_weight_gradient_route(fam, G, H, gpu_lib):
    """ReLU-KAN phases / GRAM coefficients: with an input gradient to compute, its launch accumulates them from the same G tiles
    (kan_conv_bwd_data_params); without one (a model's first layer) the weight-gradient kernel runs on the parameter-derivative planes.
    Same numbers either way."""
    torch.manual_seed(3 + H)
    C, O, B = 6 * G, 16 * G, 5
    if fam == "relu":
        layer = K.ReLUKANConv2DLayer(C, O, 3, padding=1, groups=G, base_activation=nn.SiLU).cuda()
        with torch.no_grad():
            layer.phase_low.add_(0.05 * torch.randn_like(layer.phase_low)); layer.phase_high.add_(0.05 * torch.randn_like(layer.phase_high))
        names = ("phase_low", "phase_high")
    else:
        layer = K.GRAMKANConv2DLayer(C, O, 3, padding=1, groups=G, degree=4).cuda()
        with torch.no_grad():
            layer.beta_weights.normal_(0.0, 0.2)
        names = ("beta_weights",)
    x = torch.randn(B, C, H, H, device="cuda")
    go = torch.randn(B, O, H, H, device="cuda")
    grads = []
    for need_x in (True, False):
        layer.zero_grad(set_to_none=True)
        layer(x.clone().requires_grad_(need_x)).backward(go)
        grads.append({n: getattr(layer, n).grad.clone() for n in names})
    for n in names:
        assert float(grads[1][n].abs().max()) > 0 and relerr(grads[0][n], grads[1][n]) <= 2e-5, (n, relerr(grads[0][n], grads[1][n]))
