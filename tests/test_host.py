"""CPU: host-side logic -- the C-ABI library loads and exports every symbol the header declares, plan arithmetic,
constructor / state_dict / factory parity with the reference surface, and loud failure without a GPU."""
import ctypes
import inspect
import os
import re

import numpy as np
import pytest
import torch
import torch.nn as nn

import convkan_amd as K
from convkan_amd import _lib as L
from convkan_amd import ops
from conftest import ROOT, golden_cases, load_golden
from helpers import build_layer


def test_library_exports_every_declared_symbol():
    K.build_library()
    lib = L.load()
    header = open(os.path.join(ROOT, "include", "kanconv.h")).read()
    declared = set(re.findall(r"\b(kan_[a-z_]+)\s*\(", header))
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    raw = ctypes.CDLL(L.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert b"kanconv" in lib.kan_version()


def test_struct_layouts_match_header():
    assert ctypes.sizeof(L.KanGeom) == 16 * 4 + 16              # 16 ints (incl. groups), 2 x int64
    assert L.KanGeom.groups.offset == 60 and L.KanGeom.x_bstride.offset == 64
    assert ctypes.sizeof(L.KanBasis) == 4 * 4 + 2 * 4 + 32 * 4 + 8 and L.KanBasis.chan_table.offset == 152      # + the phase-table pointer
    assert ctypes.sizeof(L.KanPlan) == 22 * 4 + 6 * 8           # 22 ints (incl. kernel-variant and expanded-copy flags), 6 x int64
    assert ctypes.sizeof(L.KanWavGeom) == 16 * 4 + 16 and L.KanWavGeom.wavelet.offset == 60 and L.KanWavGeom.x_bstride.offset == 64


def _plan(C, O, H, k=3, p=1, s=1, B=4, kind=L.BASIS_BSPLINE, nb=8, order=3, act=L.ACT_SILU, table=None):
    if table is None:
        table = tuple(float(v) for v in torch.linspace(-2.2, 2.2, 12).tolist()) if kind == L.BASIS_BSPLINE else ()
    spec = ops.ConvSpec(kind=kind, n_basis=nb, order=order, act=act, p0=4 / 7, p1=0.0, table=table, kernel=(k, k), stride=(s, s),
                        padding=(p, p), dilation=(1, 1))
    return ops._plan_cached(spec, B, C, H, H, O, C, O)


def test_plan_arithmetic():
    g, b, p = _plan(64, 128, 16, B=256)
    assert (p.P, p.K, p.IPC, p.KC) == (9, 64 * 9 * 9, 2, 18)
    assert p.Kpad == (64 * 9 // 2) * 18 and p.Opad == 128
    assert p.packed_weight_bytes == p.Kpad * p.Opad * 4
    assert p.fwd_slab_elems == 256 * 128 * 16 * 16 and p.bwd_weight_slab_elems == p.K * p.Opad
    assert p.fwd_splits >= 1 and p.bwd_data_splits >= 1 and p.bwd_weight_splits >= 1
    g, b, p = _plan(3, 16, 32)                      # config 1 shape: O padded to 64
    assert p.Opad == 64 and p.P == 9
    g, b, p = _plan(3, 64, 224, k=11, p=2, s=4, kind=L.BASIS_CHEBY, nb=5, act=L.ACT_NONE)
    assert (g.Ho, g.Wo) == (55, 55) and p.P == 5 and p.KC == 16 and p.IPC == 3


def test_plan_keeps_two_input_layers_off_the_pair_order():
    """Halo-shaped 3x3 layers pack their forward weights in channel-pair order (IPC 2, KC 2P) for the one-input halo kernel;
    LegendreKAN (order 0: basis on a second, pre-normalised tensor) must stay on the tap-major order and kernel."""
    leg = (1.0, 1.0, 0.0, 1.5, 0.0, -0.5, 5 / 3, 0.0, -2 / 3)
    _, _, one = _plan(16, 128, 8, kind=L.BASIS_POLY, nb=4, order=1, act=L.ACT_GELU, table=leg)
    _, _, two = _plan(16, 128, 8, kind=L.BASIS_POLY, nb=4, order=0, act=L.ACT_IDENTITY, table=leg)
    assert (one.P, one.IPC, one.KC) == (5, 2, 10)
    assert two.P == 5 and (two.IPC, two.KC) != (2, 10)


def test_plan_with_groups_scales_per_group_blocks():
    table = tuple(float(v) for v in torch.linspace(-2.2, 2.2, 12).tolist())
    mk = lambda G: ops.ConvSpec(kind=L.BASIS_BSPLINE, n_basis=8, order=3, act=L.ACT_SILU, p0=0.0, p1=0.0, table=table, kernel=(3, 3),
                                stride=(1, 1), padding=(1, 1), dilation=(1, 1), groups=G)
    g1, _, p1 = ops._plan_cached(mk(1), 8, 16, 14, 14, 24, 16, 24)
    g4, _, p4 = ops._plan_cached(mk(4), 8, 16, 14, 14, 24, 64, 96)        # 4 groups of the same per-group geometry
    assert (g1.groups, g4.groups) == (1, 4) and (p4.K, p4.Kpad, p4.Opad) == (p1.K, p1.Kpad, p1.Opad)      # per-group dims
    assert p4.packed_weight_bytes == 4 * p1.packed_weight_bytes and p4.bwd_data_weight_bytes == 4 * p1.bwd_data_weight_bytes
    assert p4.bwd_weight_slab_elems == 4 * p1.bwd_weight_slab_elems
    assert p4.fwd_slab_elems == 8 * 96 * 14 * 14 and p4.bwd_data_slab_elems == 8 * 64 * 14 * 14
    with pytest.raises(L.KanConvError, match="batch stride"):
        ops._plan_cached(mk(4), 8, 16, 14, 14, 24, 16, 96)                # x_bstride of one group only


def test_plan_rejects_bad_inputs():
    with pytest.raises(L.KanConvError, match="uniform"):
        _plan(3, 4, 8, table=(-2.2, -1.8, -1.4, -1.0, -0.6, -0.2, 0.2, 0.6, 1.0, 1.4, 1.9, 2.2))
    with pytest.raises(L.KanConvError, match="KAN_MAX_PLANES"):
        _plan(3, 4, 8, nb=16, order=3, table=tuple(np.linspace(-3, 3, 20).tolist()))
    with pytest.raises(L.KanConvError):
        _plan(3, 4, 2, k=5, p=0)                    # empty output


@pytest.mark.parametrize("name", golden_cases())
def test_state_dict_roundtrip_with_reference_keys(name):
    d = load_golden(name)
    layer = build_layer(d["cfg"])
    sd = {k[3:]: torch.from_numpy(v) for k, v in d.items() if k.startswith("sd.")}
    assert list(layer.state_dict().keys()) == list(sd.keys())            # same keys, same order
    layer.load_state_dict(sd, strict=True)
    for k, v in layer.state_dict().items():
        assert v.shape == sd[k].shape and torch.equal(v, sd[k])
    assert "grid" not in layer.state_dict()                              # kan_layers.py: plain attribute, not a buffer


def test_constructor_surface():
    l = K.KANConv2DLayer(6, 8, 3, groups=2, padding=1)
    for a in ("input_dim", "output_dim", "spline_order", "kernel_size", "padding", "stride", "dilation", "groups", "ndim", "grid_size",
              "grid_range", "base_activation", "norm_kwargs", "dropout", "input_dim_group", "output_dim_group", "base_conv",
              "spline_conv", "layer_norm", "prelus", "grid"):
        assert hasattr(l, a), a
    assert isinstance(l.base_activation, nn.GELU) and l.grid.shape == (12,) and l.input_dim_group == 3
    assert [n for n, _ in l.named_parameters()] == ["base_conv.0.weight", "base_conv.1.weight", "spline_conv.0.weight",
                                                    "spline_conv.1.weight", "prelus.0.weight", "prelus.1.weight"]
    assert l.spline_conv[0].weight.shape == (4, 3 * 8, 3, 3)
    assert isinstance(K.KANConv2DLayer(3, 4, 3, base_activation=None).base_activation, nn.Identity)
    assert K.KANConv2DLayer(3, 4, 3, norm_layer=nn.BatchNorm2d, affine=True, bogus=1).layer_norm[0].affine       # kwargs filtered
    f = K.FastKANConv2DLayer(3, 4, 3)
    assert list(f.state_dict()) == ["base_conv.0.weight", "spline_conv.0.weight", "rbf.grid"] and not f.rbf.grid.requires_grad
    c = K.ChebyKANConv2DLayer(3, 4, 3, degree=4)
    assert list(c.state_dict()) == ["arange", "poly_conv.0.weight"] and c.arange.shape == (1, 1, 5, 1, 1)
    for cls in (K.KANConv2DLayer, K.FastKANConv2DLayer, K.ChebyKANConv2DLayer):
        for bad, msg in ((dict(groups=0), "positive"), (dict(groups=2), "input_dim must be divisible")):
            with pytest.raises(ValueError, match=msg):
                cls(3, 4, 3, **bad)
        with pytest.raises(ValueError, match="output_dim must be divisible"):
            cls(4, 3, 3, groups=2)


def test_1d_shims_surface():
    """kan_layers.py:287-297, fast_kan_layers.py:151-162, cheby_kan_layers.py:134-141: Conv1d weight holders, InstanceNorm1d."""
    a = K.KANConv1DLayer(4, 6, 3, groups=2, padding=1)
    assert a.ndim == 1 and isinstance(a.base_conv[0], nn.Conv1d) and isinstance(a.layer_norm[0], nn.InstanceNorm1d)
    assert a.spline_conv[1].weight.shape == (3, 16, 3) and a.conv_spec().kernel == (1, 3) and a.conv_spec().padding == (0, 1)
    f = K.FastKANConv1DLayer(3, 4, 5, stride=2, dropout=0.1)
    assert isinstance(f.dropout, nn.Dropout1d) and f.conv_spec().stride == (1, 2) and list(f.state_dict())[-1] == "rbf.grid"
    c = K.ChebyKANConv1DLayer(3, 4, 3, degree=4)
    assert c.arange.shape == (1, 1, 5, 1) and list(c.state_dict()) == ["arange", "poly_conv.0.weight"]
    v = K.KANConv3DLayer(4, 6, 3, groups=2, padding=1, dropout=0.1)            # kan_layers.py:261-271
    assert v.spline_conv[0].weight.shape == (3, 16, 3, 3, 3) and isinstance(v.layer_norm[0], nn.InstanceNorm3d) and v.ndim == 3
    assert isinstance(v.dropout, nn.Dropout3d) and list(v.state_dict())[:2] == ["base_conv.0.weight", "base_conv.1.weight"]
    assert K.FastKANConv3DLayer(4, 6, 3).spline_conv[0].weight.shape == (6, 32, 3, 3, 3)
    assert K.ChebyKANConv3DLayer(4, 6, 3, degree=4).arange.shape == (1, 1, 5, 1, 1, 1)
    with pytest.raises(NotImplementedError):
        K.ReLUConvNDLayer(nn.Conv3d, nn.InstanceNorm3d, None, 3, 4, 3, ndim=3)  # 3-D: the three hot-path families only
    with pytest.raises(Exception):
        a(torch.zeros(2, 4, 8, 8))                                # a 1-D layer takes [B, C, L]


def test_polynomial_family_surface():
    """State-dict keys, attributes and validation messages of the recurrence families (e.g. lucas_kan_layers.py:76-139)."""
    for cls in (K.BesselKANConv2DLayer, K.FibonacciKANConv2DLayer, K.HermiteKANConv2DLayer, K.LucasKANConv2DLayer, K.TaylorKANConv2DLayer):
        m = cls(4, 6, 3, degree=3, groups=2, padding=1)
        n = 3 if cls is K.TaylorKANConv2DLayer else 4
        assert list(m.state_dict()) == ["base_conv.0.weight", "base_conv.1.weight", "poly_conv.0.weight", "poly_conv.1.weight",
                                        "prelus.0.weight", "prelus.1.weight"]
        assert m.poly_conv[0].weight.shape == (3, 2 * n, 3, 3) and m.poly_input_dim_group == 2 * n
        with pytest.raises(ValueError, match="positive"):
            cls(4, 6, 3, degree=3, groups=0)
    with pytest.raises(ValueError, match="at least 1"):
        K.FibonacciKANConv2DLayer(3, 4, 3, degree=0)
    with pytest.raises(ValueError, match="at least 1"):
        K.TaylorKANConv2DLayer(3, 4, 3, degree=0)
    with pytest.raises(ValueError, match="non-negative"):
        K.LucasKANConv2DLayer(3, 4, 3, degree=-1)
    with pytest.raises(ValueError, match="greater than -0.5"):
        K.GegenbauerKANConv2DLayer(3, 4, 3, degree=2, alpha_param=-0.5)
    with pytest.raises(ValueError, match="greater than -1"):
        K.LaguerreKANConv2DLayer(3, 4, 3, degree=2, alpha=-1.0)
    j = K.JacobiKANConv2DLayer(4, 6, 3, degree=3, groups=2)
    assert list(j.state_dict()) == ["poly_weights", "base_conv.0.weight", "base_conv.1.weight"] and j.poly_weights.shape == (2, 3, 8, 3, 3)
    spec = K.LaguerreKANConv2DLayer(3, 4, 3, degree=3, alpha=1.0).conv_spec()
    assert spec.kind == L.BASIS_POLY and spec.n_basis == 4 and spec.order == 1
    assert spec.table[:3] == (1.0, -1.0, 2.0) and spec.table[3:6] == (-0.5, 2.0, -1.0)      # L_2 = ((4 - t) L_1 - 2 L_0) / 2


def test_factory_signatures_and_same_padding():
    F = K.CONV_KAN_FACTORY
    poly = {"BesselKAN", "FibonacciKAN", "GegenbauerKAN", "HermiteKAN", "JacobiKAN", "LaguerreKAN", "LucasKAN", "TaylorKAN"}
    assert set(F) == {"KAN", "FastKAN", "ChebyKAN", "FourierKAN", "LegendreKAN", "BersnsteinKAN", "ReLUKAN", "GRAMKAN", "WavKAN", "conv"} | poly      # kan_conv.py:726-745: all 18
    wv = F["WavKAN"](4, 6, 3, groups=2)                          # kan_conv.py:278-318: 'fast' wavelet conv, InstanceNorm2d, same padding
    assert wv.padding == 1 and wv.wavelet_type == "mexican_hat" and type(wv.wavelet_conv[0]).__name__ == "WaveletConvNDFast"
    assert isinstance(wv.layer_norm[0], nn.InstanceNorm2d) and isinstance(wv.base_activation, nn.SiLU)
    assert list(wv.state_dict()) == ["base_conv.0.weight", "base_conv.1.weight", "wavelet_conv.0.scale", "wavelet_conv.0.translation",
                                     "wavelet_conv.0.wavelet_weights.weight", "wavelet_conv.0.wavelet_out.weight", "wavelet_conv.1.scale",
                                     "wavelet_conv.1.translation", "wavelet_conv.1.wavelet_weights.weight", "wavelet_conv.1.wavelet_out.weight"]
    assert wv.wavelet_conv[0].scale.shape == (1, 3, 2, 1, 1) and wv.wavelet_conv[0].wavelet_weights.weight.shape == (3, 2, 3, 3)
    assert K.WavKANConv2DLayer(4, 6, 3, wav_version="fast_plus_one").wavelet_conv[0].wavelet_weights.weight.shape == (6, 1, 4, 3, 3)
    assert len(K.WavKANConv2DLayer(4, 6, 3, wav_version="base").wavelet_conv[0].wavelet_weights) == 6
    assert isinstance(K.WavKANConv2DLayer(4, 6, 3).layer_norm[0], nn.BatchNorm2d)       # the class default (wav_kan_layers.py:467)
    gr = F["GRAMKAN"](4, 6, 3, groups=2, dilation=2)            # kan_conv.py:158-194; gram_kan_layers.py:85-148
    assert (gr.padding, gr.dilation, gr.degree) == (2, 2, 3) and isinstance(gr.base_activation, nn.SiLU)
    assert list(gr.state_dict()) == ["poly_weights", "beta_weights", "base_conv.0.weight", "base_conv.1.weight"]
    assert gr.poly_weights.shape == (2, 3, 2 * 4, 3, 3) and gr.beta_weights.shape == (4,) and gr.conv_spec().kind == L.BASIS_GRAM
    assert list(inspect.signature(F["GRAMKAN"]).parameters)[3] == "degree" and "base_activation" not in inspect.signature(F["GRAMKAN"]).parameters
    assert abs(float(gr._beta_factor[2]) - 3 * 1 * 1 / (4 / 3.0)) < 1e-6 and float(gr._beta_factor[:2].abs().sum()) == 0
    rl = F["ReLUKAN"](4, 6, 3, groups=2, dilation=2)             # kan_conv.py:652-690; relu_kan_layers.py:41-116
    assert (rl.padding, rl.dilation, rl.g, rl.k, rl.r, rl.train_ab) == (2, 1, 5, 3, 6.25, True) and isinstance(rl.base_activation, nn.GELU)
    assert list(rl.state_dict()) == ["phase_low", "phase_high", "base_conv.0.weight", "base_conv.1.weight", "relukan_conv.0.weight",
                                     "relukan_conv.1.weight"]
    assert rl.phase_low.shape == (1, 2, 8, 1, 1) and rl.relukan_conv[0].weight.shape == (3, 16, 3, 3)
    assert torch.equal(rl.phase_low[0, 1, :, 0, 0], torch.arange(-3, 5) / 5) and torch.equal(rl.phase_high, rl.phase_low + 4 / 5)
    assert not F["ReLUKAN"](4, 6, 3, train_ab=False).phase_low.requires_grad
    assert isinstance(K.ReLUKANConv2DLayer(4, 6, 3).base_activation, nn.SiLU) and K.ReLUKANConv1DLayer(4, 6, 3).phase_high.shape == (1, 4, 8, 1)
    assert isinstance(K.ReLUKANConv1DLayer(4, 6, 3, base_activation=nn.GELU).base_activation, nn.SiLU)       # dropped, as in the reference
    assert K.ReLUKANConv2DLayer(4, 6, 3).conv_spec().kind == L.BASIS_RELU
    lg = F["LegendreKAN"](4, 6, 3, dilation=2)
    assert (lg.padding, lg.dilation, lg.degree) == (2, 2, 3) and isinstance(lg.base_activation, nn.SiLU)
    assert list(inspect.signature(F["LegendreKAN"]).parameters)[3] == "degree"
    bs = F["BersnsteinKAN"](4, 6, 3, groups=2)
    assert list(bs.state_dict()) == ["poly_weights", "base_conv.0.weight", "base_conv.1.weight"] and (bs.inputdim, bs.outdim) == (4, 6)
    assert inspect.signature(F["FourierKAN"]).parameters["grid_size"].default == 3
    fl = F["FourierKAN"](4, 6, 3, groups=2)
    assert list(fl.state_dict())[:4] == ["base_conv.0.weight", "base_conv.1.weight", "fourier_conv.0.weight", "fourier_conv.1.weight"]
    assert fl.fourier_conv[0].weight.shape == (3, 2 * 6, 3, 3) and fl.fourier_input_dim_group == 12
    for name in poly:                                            # kan_conv.py:354-724: shared leading arguments and defaults
        sig = inspect.signature(F[name])
        assert list(sig.parameters)[:7] == ["in_planes", "out_planes", "kernel_size", "groups", "stride", "dilation", "padding"]
        assert sig.parameters["degree"].default == 3 and sig.parameters["base_activation"].default is nn.GELU
        layer = F[name](4, 6, 3, dilation=2)                     # dilation only enters the 'same' padding, as in the reference
        assert layer.padding == 2 and layer.dilation == 1 and layer.degree == 3
    assert inspect.signature(F["GegenbauerKAN"]).parameters["alpha_param"].default == 0.0
    assert inspect.signature(F["LaguerreKAN"]).parameters["alpha"].default == 1.0
    assert (F["JacobiKAN"](4, 6, 3).a, F["JacobiKAN"](4, 6, 3, b=0.5).b) == (1.0, 0.5)
    sig = inspect.signature(F["KAN"])
    assert list(sig.parameters)[:3] == ["in_planes", "out_planes", "kernel_size"]
    assert sig.parameters["grid_size"].default == 5 and sig.parameters["base_activation"].default is nn.GELU
    assert inspect.signature(F["FastKAN"]).parameters["grid_range"].default == [-2, 2]
    assert inspect.signature(F["ChebyKAN"]).parameters["degree"].default == 3
    assert F["KAN"](3, 8, 3).padding == 1 and F["KAN"](3, 8, 5, dilation=2).padding == 4
    assert F["KAN"](3, 8, (3, 5)).padding == (1, 2)
    assert F["FastKAN"](3, 8, 3, l1_decay=0.0).padding == 1 and F["ChebyKAN"](3, 8, 3, degree=4, affine=True).layer_norm[0].affine
    wrapped = F["KAN"](3, 8, 3, l1_decay=0.1)                    # kan_conv.py:66-68: L1(conv, l1_decay)
    assert type(wrapped).__name__ == "L1" and isinstance(wrapped.module, K.KANConv2DLayer) and wrapped.weight_decay == 0.1
    assert list(wrapped.state_dict())[0] == "module.base_conv.0.weight"
    assert type(F["LucasKAN"](4, 6, 3, l1_decay=0.01)).__name__ == "L1" and type(F["conv"](3, 8, 3, l1_decay=0.1)[0]).__name__ == "L1"
    # a base_activation without a device functor is applied by the host (identity functor + two-input kernels) where the
    # layer has that form -- ReLU-KAN included since round 3 (relu_kan_layers.py:57 takes any module).  GRAM passes its PLANES through the
    # activation inside the kernel (gram_kan_layers.py:181), so only the device functors serve it; its 1-D / 2-D / 3-D classes fix SiLU anyway (:203-231)
    soft = K.KANConv2DLayer(3, 4, 3, base_activation=nn.Softplus)
    assert soft.conv_spec().act == L.ACT_IDENTITY and isinstance(soft.base_activation, nn.Softplus)
    rsoft = K.ReLUKANConv2DLayer(3, 4, 3, base_activation=nn.Softplus)
    assert rsoft.conv_spec().act == L.ACT_IDENTITY and isinstance(rsoft.base_activation, nn.Softplus)


def test_no_cpu_fallback():
    layer = K.KANConv2DLayer(3, 4, 3, padding=1)
    with pytest.raises(L.KanConvError, match="no CPU fallback"):
        layer(torch.randn(1, 3, 8, 8))


def test_models_build_with_reference_parameter_names():
    import json
    from convkan_amd.models import alexnet_kan, cfgs, vggkan
    assert cfgs["VGG11"] == [64, "M", 128, "M", 256, 256, "M", 512, 512, "M", 512, 512]
    small = vggkan(3, 10, arch="VGG16_kansmall", kan_conv="KAN", classifier_type="Linear")
    assert small.name == "VGGKAN_Linear_KAN_VGG16_kansmall"
    with pytest.raises(ValueError, match="Unknown arch"):
        vggkan(3, 10, arch="VGG13")
    d = np.load(os.path.join(ROOT, "tests", "golden", "model_kan_vgg11.npz"))
    names = json.loads(bytes(d["names"]).decode())
    m = vggkan(3, 10, arch="VGG11", kan_conv="KAN", classifier_type="Linear")
    assert [n for n, _ in m.named_parameters()] == names and sum(p.numel() for p in m.parameters()) == 82964690
    a = alexnet_kan(num_classes=10, kan_conv="ChebyKAN", degree=4, arch="small")
    assert a.features[0].kernel_size == 5 and a.features[0].layer_norm[0].affine
    # the reference's 'KAN' head (kan_alexnet.py:184-199): two Linear+ReLU stages, the last stage a B-spline MLP KAN `kan_fc3`;
    # names / shapes as the reference builds them (checked against the imported reference when this head was added)
    k = alexnet_kan(num_classes=10, kan_conv="KAN", classifier_type="KAN", arch="small", grid_size=4)
    assert k.name == "AlexNet_KAN_KAN_KAN"
    sd = {n: tuple(v.shape) for n, v in k.state_dict().items() if "kan_fc3" in n}
    assert sd == {"classifier.kan_fc3.layers.0.base_weight": (10, 1024), "classifier.kan_fc3.layers.0.spline_weight": (10, 1024, 7),
                  "classifier.kan_fc3.layers.0.layer_norm.weight": (10,), "classifier.kan_fc3.layers.0.layer_norm.bias": (10,),
                  "classifier.kan_fc3.layers.0.prelu.weight": (1,)}
    assert "classifier.fc3.weight" not in k.state_dict() and "classifier.fc2.weight" in k.state_dict()
    assert alexnet_kan(num_classes=10, classifier_type="AlexNetKAN", arch="small").name == "AlexNet_AlexNetKAN_KAN_KAN"     # plain head, as the reference


def test_l1_l2_wrapper_hook_semantics():
    """utils/regularization.py:57-159: a full backward hook on the wrapped module that seeds .grad with the penalty for every
    selected parameter whose gradient is still None or all zeros when it fires, and leaves real gradients alone."""
    from convkan_amd.utils import L1, L2
    torch.manual_seed(0)
    lin = nn.Linear(5, 3)
    for cls, pen in ((L1, lambda p: 0.1 * torch.sign(p)), (L2, lambda p: 0.1 * p)):
        w = cls(lin, 0.1)
        assert len(lin._backward_hooks) == 1 and list(w.state_dict()) == ["module.weight", "module.bias"]
        lin.zero_grad(set_to_none=True)
        w._weight_decay_hook()                                               # gradients still None
        assert all(torch.equal(p.grad, pen(p.detach())) for p in lin.parameters())
        lin.weight.grad = torch.zeros_like(lin.weight); lin.bias.grad = torch.ones_like(lin.bias)
        w._weight_decay_hook()                                               # all-zero -> penalty; non-zero -> untouched
        assert torch.equal(lin.weight.grad, pen(lin.weight.detach())) and torch.equal(lin.bias.grad, torch.ones(3))
        w.remove()
        assert len(lin._backward_hooks) == 0
    lin.zero_grad(set_to_none=True)
    only_w = L1(lin, 0.5, name="weight")
    only_w._weight_decay_hook()
    assert lin.bias.grad is None and torch.equal(lin.weight.grad, 0.5 * torch.sign(lin.weight.detach()))
    x = torch.randn(4, 5, requires_grad=True)
    assert torch.equal(only_w(x), lin(x))
    only_w(x).sum().backward()                                               # the hook is live in a real backward pass
    assert lin.bias.grad is not None
    with pytest.raises(ValueError):
        L1(lin, -1.0)


def test_image_runs_never_exceed_the_launch_limit(monkeypatch):
    """ops._image_runs: a batch above the 2 GiB launch limit is cut into the fewest equal-ish runs of whole images, and no run may itself
    exceed the limit (B = 511 images of 8 MiB used to give 2 runs of 256 = 2^31 bytes, which kan_plan then rejected)."""
    spec = K.KANConv2DLayer(4, 4, 3, padding=1).conv_spec()

    class _X:                                                    # shape only: nothing is allocated
        def __init__(self, *shape):
            self.shape = shape
    for B, C, H, W, O in [(511, 64, 256, 128, 4), (512, 64, 256, 128, 4), (7, 6, 12, 12, 10), (1000, 3, 300, 301, 64), (257, 1, 2048, 1024, 1),
                          (3, 8, 4096, 4096, 8), (4, 8, 4096, 4096, 8), (255, 64, 256, 128, 64)]:
        Ho, Wo = spec.out_hw(H, W)
        per_image = 4 * max(C * H * W, O * Ho * Wo)
        n = ops._image_runs(spec, _X(B, C, H, W), O)
        if B * per_image <= ops.MAX_TENSOR_BYTES:
            assert n is None
            continue
        assert n is not None and 1 <= n and n * per_image <= ops.MAX_TENSOR_BYTES, (B, n, per_image)
        runs = -(-B // n)
        assert runs == -(-B // (ops.MAX_TENSOR_BYTES // per_image)), (B, n, runs)        # the fewest runs that fit
        assert n - (B - (runs - 1) * n) < runs + n // 2 or runs == 1                      # equal-ish: the last run is not a sliver by construction
    monkeypatch.setattr(ops, "MAX_TENSOR_BYTES", 4 * 4 * 8 * 8 - 1)                        # one image above the limit raises
    with pytest.raises(L.KanConvError):
        ops._image_runs(spec, _X(2, 4, 8, 8), 4)


def test_taylor_3d_shim_sizes_its_basis_like_the_2d_layer():
    """TaylorKANConv3DLayer holds `degree` planes (taylor_kan_layers.py compute_taylor_basis), not degree + 1: the shared 3-D forward must
    build its basis from the same count as conv_spec() (it used degree + 1 and indexed past the coefficient list)."""
    lay = K.TaylorKANConv3DLayer(2, 4, 3, degree=3, padding=1)
    assert lay.poly_conv[0].weight.shape[1] == 2 * 3 and lay.conv_spec().n_basis == 3
    seen = {}
    import convkan_amd.layers.poly_layers as PL
    orig = PL.conv3d_stage
    try:
        PL.conv3d_stage = lambda kw, *a, **k: seen.update(kw) or (_ for _ in ()).throw(RuntimeError("stop"))
        with pytest.raises(RuntimeError, match="stop"):
            lay._forward3d(torch.randn(1, 2, 4, 4, 4))
    finally:
        PL.conv3d_stage = orig
    assert seen["n_basis"] == 3 and len(seen["table"]) == len(lay.conv_spec().table)


def test_reference_model_files_build_unchanged_on_this_factory():
    """tests/golden/dropin_check.json is written in the build container by `make_golden.py --dropin-only`: the reference's UNMODIFIED
    models/kan_vgg.py / kan_alexnet.py, with CONV_KAN_FACTORY swapped for this repo's, gave the reference's own state_dict keys / shapes
    and a strict load_state_dict round trip both ways (north_star: "drops into kan_vgg.py / kan_alexnet.py unchanged").  The reference
    cannot travel, so here the record is checked against what this repo's layers and model counterparts build TODAY: a layer change
    that alters a key or a shape fails this test until the drop-in check has been re-run against the reference."""
    import json
    from convkan_amd.models import alexnet_kan, vggkan
    with open(os.path.join(ROOT, "tests", "golden", "dropin_check.json")) as f:
        rec = json.load(f)
    assert set(rec["reference_files"]) == {"models/kan_vgg.py", "models/kan_alexnet.py", "layers/kan_conv.py"}
    build = {"vgg11_kan_linear": lambda: vggkan(3, 10, arch="VGG11", kan_conv="KAN", classifier_type="Linear"),
             "vgg11_kan_kanhead": lambda: vggkan(3, 10, arch="VGG11", kan_conv="KAN", classifier_type="KAN"),
             "vgg16small_fastkan": lambda: vggkan(3, 10, arch="VGG16_small", kan_conv="FastKAN", classifier_type="Linear"),
             "alexnet_chebykan_deg4": lambda: alexnet_kan(num_classes=10, kan_conv="ChebyKAN", degree=4),
             "alexnet_kan_kanhead_small": lambda: alexnet_kan(num_classes=10, kan_conv="KAN", classifier_type="KAN", arch="small", grid_size=4)}
    assert set(rec["models"]) == set(build)
    for name, mk in build.items():
        r = rec["models"][name]
        assert r["strict_round_trip"] is True and r["hip_layers_built_by_reference_code"] > 0
        m = mk()
        assert [(k, list(v.shape)) for k, v in m.state_dict().items()] == [(k, list(s)) for k, s in r["state_dict"]], name
        assert sum(p.numel() for p in m.parameters()) == r["parameters"] and getattr(m, "name", type(m).__name__) == r["class_name"], name


def test_plane_windows_cover_every_basis_of_a_wide_grid():
    """grid_size + spline_order + 1 > KAN_MAX_PLANES = 16 (the reference takes any grid_size, kan_layers.py:117-131): the layer cuts its bases
    into windows of <= 16 planes, each a B-spline spec on the matching slice of the fp32 knots (basis j lives on knots j .. j + order + 1)."""
    for gs, order, act in [(16, 3, nn.SiLU), (40, 2, nn.GELU), (13, 3, None), (12, 3, nn.SiLU), (100, 1, nn.SiLU)]:
        lay = K.KANConv2DLayer(3, 4, 3, padding=1, grid_size=gs, spline_order=order, base_activation=act)
        n, knots = gs + order, [float(v) for v in lay.grid.tolist()]
        wins = lay._plane_windows()
        assert wins[0][1] == 0 and wins[-1][2] == n and all(a[2] == b[1] for a, b in zip(wins, wins[1:]))      # contiguous cover of [0, n)
        for i, (spec, j0, j1, has_base) in enumerate(wins):
            assert has_base == (i == 0) and spec.n_basis == j1 - j0 and spec.n_basis + int(spec.has_base) <= L.KAN_MAX_PLANES
            assert spec.table == tuple(knots[j0:j1 + order + 1]) and spec.order == order
            assert spec.has_base == has_base                         # (base_activation=None is Identity: still a base branch, kan_layers.py:132)
        if n + 1 <= L.KAN_MAX_PLANES:
            assert len(wins) == 1


def test_alexnet_plain_pool_detection():
    """models/kan_alexnet.py fuses only square, unpadded, undilated floor-mode pools (the reference's MaxPool2d(kernel_size=3, stride=2), kan_alexnet.py:120-126)."""
    import torch.nn as nn
    from convkan_amd.models.kan_alexnet import _plain_pool
    assert _plain_pool(nn.MaxPool2d(kernel_size=3, stride=2)) == (3, 2)
    assert _plain_pool(nn.MaxPool2d(2)) == (2, 2) and _plain_pool(nn.MaxPool2d((3, 3), (1, 1))) == (3, 1)
    for bad in (nn.MaxPool2d(3, 2, padding=1), nn.MaxPool2d(3, 2, ceil_mode=True), nn.MaxPool2d((3, 2), 2), nn.MaxPool2d(3, 2, dilation=2),
                nn.MaxPool2d(2, 3), nn.MaxPool2d(3, 2, return_indices=True)):
        assert _plain_pool(bad) is None


def test_opt_in_switches_restore_their_flags_and_pool_arguments_are_checked():
    """ops.always_pack / ops.split_precision_inference are context managers that leave no state behind (also on an exception); the fused-pool argument
    takes True, False or a (kernel, stride) pair with 2 <= kernel <= 15, 1 <= stride <= kernel."""
    from convkan_amd import ops
    assert ops._ALWAYS_PACK is False and ops._SPLIT_INFERENCE is False
    with ops.always_pack():
        assert ops._ALWAYS_PACK is True
        with ops.split_precision_inference():
            assert ops._SPLIT_INFERENCE is True
        assert ops._SPLIT_INFERENCE is False
    assert ops._ALWAYS_PACK is False
    with pytest.raises(RuntimeError):
        with ops.split_precision_inference():
            raise RuntimeError("x")
    assert ops._SPLIT_INFERENCE is False
    assert ops._norm_pool(False) is False and ops._norm_pool(True) is True and ops._norm_pool((3, 2)) == (3, 2) and ops._norm_pool([2, 2]) == (2, 2)
    for bad in ((1, 1), (3, 4), (16, 2), (3, 0)):
        with pytest.raises(L.KanConvError):
            ops._norm_pool(bad)
