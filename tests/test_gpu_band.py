"""Band kernels (csrc/kan_direct.hip): layers of few input channels and layers whose output count fills no 128-wide tile, any kernel size /
stride / dilation / padding -- forward against the oracle (calibrated tolerance rule, helpers.check_vs_oracle), and the plan flag that says
the band kernel ran.  Shapes: the first layers of BASELINE.json's configs (KAN-VGG11 3->64@32x32, FastKAN 3->64 with padding 0, ChebyKAN-AlexNet
3->64 k11 s4 and 64->192 k5), plus strides / dilations / rectangular kernels / odd channel counts / groups / ragged tiles."""
import pytest
import torch
import torch.nn as nn

import convkan_amd as K
from convkan_amd import ops
from helpers import check_vs_oracle
from test_gpu_oracle import _cfg

pytestmark = pytest.mark.gpu

CASES = [
    # (id, kind, C, O, H, W, B, layer kwargs, cfg kwargs)
    ("vgg_l0", "bspline", 3, 64, 32, 32, 8, dict(kernel_size=3, padding=1, base_activation=nn.SiLU), dict(act="silu")),
    ("vgg_l0_gelu_b3", "bspline", 3, 64, 32, 32, 3, dict(kernel_size=3, padding=1), dict(act="gelu")),
    ("fastkan_p0", "rbf", 3, 64, 32, 32, 16, dict(kernel_size=3), dict(p=0)),
    ("fastkan_p1_o128", "rbf", 3, 128, 16, 16, 5, dict(kernel_size=3, padding=1), dict()),
    ("alex_l0_k11s4", "cheby", 3, 64, 224, 224, 2, dict(kernel_size=11, degree=4, stride=4, padding=2, affine=True), dict(k=11, s=4, p=2, degree=4)),
    ("alex_l1_k5_o192", "cheby", 64, 192, 27, 27, 3, dict(kernel_size=5, degree=4, padding=2, affine=True), dict(k=5, p=2, degree=4)),
    ("mnist_1ch", "bspline", 1, 32, 28, 28, 7, dict(kernel_size=3, padding=1, base_activation=nn.SiLU), dict(act="silu")),
    ("stride2_k3", "bspline", 3, 32, 33, 31, 4, dict(kernel_size=3, stride=2, padding=1, base_activation=nn.SiLU), dict(s=2, act="silu")),
    ("stride3_k5_d2", "cheby", 2, 64, 29, 40, 3, dict(kernel_size=5, degree=4, stride=3, padding=3, dilation=2), dict(k=5, s=3, p=3, d=2, degree=4)),
    ("k1_s2", "bspline", 3, 64, 9, 9, 5, dict(kernel_size=1, stride=2, padding=0, base_activation=nn.SiLU), dict(k=1, s=2, p=0, act="silu")),
    ("stride_gt_kernel", "bspline", 2, 64, 17, 17, 2, dict(kernel_size=2, stride=3, padding=0, base_activation=nn.SiLU), dict(k=2, s=3, p=0, act="silu")),
    ("rect_k3x1", "bspline", 3, 64, 12, 7, 3, dict(kernel_size=(3, 1), padding=(1, 0), base_activation=nn.SiLU), dict(k=(3, 1), p=(1, 0), act="silu")),
    ("odd_c5_o40", "cheby", 5, 40, 11, 13, 4, dict(kernel_size=3, degree=4, padding=1), dict(degree=4)),
    ("c6_o192_two_splits", "cheby", 6, 192, 14, 14, 6, dict(kernel_size=3, degree=3, padding=1), dict(degree=3)),
    ("groups2_c6_o128", "bspline", 6, 128, 10, 10, 3, dict(kernel_size=3, padding=1, groups=2, base_activation=nn.SiLU), dict(groups=2, act="silu")),
    ("tiny_plane_many_images", "bspline", 3, 64, 3, 3, 40, dict(kernel_size=3, padding=1, base_activation=nn.SiLU), dict(act="silu")),
    ("pad_past_kernel", "bspline", 2, 64, 5, 6, 3, dict(kernel_size=3, padding=4, base_activation=nn.SiLU), dict(p=4, act="silu")),
    ("lucas_deg4_c3", "lucas", 3, 64, 16, 16, 4, dict(kernel_size=3, degree=3, padding=1, base_activation=nn.SiLU), dict(degree=3, act="silu")),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_band_forward_vs_oracle(case, gpu_lib):
    name, kind, C, O, H, W, B, kw, ckw = case
    torch.manual_seed(len(name) + C + O)
    cls = {"bspline": K.KANConv2DLayer, "rbf": K.FastKANConv2DLayer, "cheby": K.ChebyKANConv2DLayer, "lucas": K.LucasKANConv2DLayer}[kind]
    kw = dict(kw)
    ks = kw.pop("kernel_size")
    layer = cls(C, O, ks, **kw)
    G = kw.get("groups", 1)
    plan = ops._plan_cached(layer.conv_spec(), B, C // G, H, W, O // G, C, O)[2]
    assert plan.fwd_band == 1, f"{name}: the plan does not route this layer to the band forward kernel"
    # the weight gradient takes the band kernel too (packed gradient in band order) where a compile-time spec exists for it, except on small
    # padded planes with >= 16 images, which keep the position-major tap-skipping launch
    want_bw = 1 if (kind in ("bspline", "rbf", "cheby") and kw.get("degree", 4) == 4 and name != "tiny_plane_many_images") else 0
    assert plan.bwd_weight_band == want_bw, (name, plan.bwd_weight_band)
    cfg = _cfg(kind, C, O, **{"k": ks, **ckw})
    check_vs_oracle(layer, cfg, torch.randn(B, C, H, W) * 1.3, groups=G, tag=name)


def test_band_forward_is_deterministic_and_batch_independent(gpu_lib):
    """Pixel tiles are 128 consecutive pixels in (image, row, column) order, so images share tiles on small or odd planes: an image's result
    must not depend on its neighbours in the batch, and two runs must agree bit for bit."""
    torch.manual_seed(4)
    layer = K.ChebyKANConv2DLayer(3, 64, 5, degree=4, stride=2, padding=2).cuda()
    x = torch.randn(9, 3, 13, 11, device="cuda")
    spec = layer.conv_spec()
    assert ops._plan_cached(spec, 9, 3, 13, 11, 64, 3, 64)[2].fwd_band == 1
    w = [m.weight for m in layer.poly_conv]
    z1 = ops.kan_conv(spec, x, None, [], w)
    z2 = ops.kan_conv(spec, x, None, [], w)
    assert torch.equal(z1, z2)
    for i in (0, 4, 8):
        zi = ops.kan_conv(spec, x[i:i + 1].contiguous(), None, [], w)
        assert torch.equal(zi[0], z1[i]), i


def test_band_forward_nan_stays_local(gpu_lib):
    """The odd plane count of a 3-channel B-spline step (27) is padded with a zero-weight row that reads a zero LDS word, never a neighbour's
    value: a NaN input pixel poisons only the outputs whose receptive field holds it."""
    torch.manual_seed(5)
    layer = K.KANConv2DLayer(3, 64, 3, padding=1, base_activation=nn.SiLU).cuda()
    x = torch.randn(2, 3, 16, 16, device="cuda")
    x[0, 1, 5, 7] = float("nan")
    z = ops.kan_conv(layer.conv_spec(), x, None, [m.weight for m in layer.base_conv], [m.weight for m in layer.spline_conv])
    bad = torch.isnan(z).any(dim=1)                             # [B, H, W]
    want = torch.zeros_like(bad)
    want[0, 4:7, 6:9] = True
    assert torch.equal(bad, want)
