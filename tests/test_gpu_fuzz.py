"""Seeded random geometries through the HIP path vs the CPU oracle (forward, input gradient, every parameter gradient).

Shapes are drawn so that every kernel family gets hit: 64/128/256-output tiles, halo and tap-major forward, position-major
small planes, grouped and depthwise launches, strides / dilations / paddings / rectangular kernels, 1..300 images."""
import random

import pytest
import torch
import torch.nn as nn

import convkan_amd as K
from helpers import check_vs_oracle
from test_gpu_oracle import _cfg

pytestmark = pytest.mark.gpu

FAMILIES = ["KAN", "KAN", "KAN", "FastKAN", "ChebyKAN", "LucasKAN", "FourierKAN", "JacobiKAN"]
KIND = {"KAN": "bspline", "FastKAN": "rbf", "ChebyKAN": "cheby", "LucasKAN": "lucas", "FourierKAN": "fourier", "JacobiKAN": "jacobi"}


def _draw(seed):
    r = random.Random(seed)
    fam = r.choice(FAMILIES)
    groups = r.choice([1, 1, 1, 2, 4, "dw"])
    cg = r.choice([1, 2, 3, 4, 6, 8, 16]) if groups != "dw" else 1
    og = r.choice([1, 2, 5, 16, 64, 128]) if groups != "dw" else r.choice([1, 2])
    if r.random() < 0.35 and groups != "dw":                  # halo / big-tile friendly draws
        cg, og = r.choice([2, 4, 8, 16]), r.choice([128, 256])
    G = r.choice([3, 5, 8]) if groups == "dw" else groups
    C, O = cg * G, og * G
    if fam == "JacobiKAN":
        k = r.choice([1, 3])
        kk = k
    else:
        kk = r.choice([1, 3, 3, 3, 5, (3, 1), (1, 3)]) if fam not in ("ChebyKAN",) else r.choice([1, 3, 3, 5])
    kh, kw = (kk, kk) if isinstance(kk, int) else kk
    s = r.choice([1, 1, 1, 2])
    d = r.choice([1, 1, 2])
    p = r.choice([0, 1, 1, 2])
    H, W = r.choice([(2, 2), (4, 4), (8, 8), (16, 16), (32, 32), (5, 7), (9, 4), (12, 12), (6, 6)])
    if (H + 2 * p - d * (kh - 1) - 1) // s + 1 <= 0 or (W + 2 * p - d * (kw - 1) - 1) // s + 1 <= 0:
        p = d * (max(kh, kw) - 1)                             # make the output non-empty
    ho, wo = (H + 2 * p - d * (kh - 1) - 1) // s + 1, (W + 2 * p - d * (kw - 1) - 1) // s + 1
    if ho * wo < 4:                                           # InstanceNorm over 1-3 values: torch refuses 1, and the gradient
        s, d = 1, 1                                           # through 2-3 values is pure cancellation even in the reference
        p = max(p, max(kh, kw) // 2)
    B = r.choice([1, 2, 3, 8, 17, 40]) if O * H * W <= 4096 * 16 else r.choice([1, 2, 5])
    return dict(fam=fam, C=C, O=O, G=G, k=kk, s=s, d=d, p=p, H=H, W=W, B=B)


def _check(layer, cfg, x, c):
    """max(stated, 4 x the fp32 oracle's own distance from fp64) per tensor (helpers.check_vs_oracle).  KAN_FUZZ_SCALE widens the stated
    part for exploration; the committed value is 1 and no shape class gets a multiplier of its own."""
    check_vs_oracle(layer, cfg, x, groups=c["G"], tag=c, scale=float(__import__("os").environ.get("KAN_FUZZ_SCALE", "1")))


_OFF = int(__import__("os").environ.get("KAN_FUZZ_OFFSET", "0"))      # KAN_FUZZ_OFFSET / _N / _FAM_N: explore other seed ranges


@pytest.mark.parametrize("seed", range(_OFF, _OFF + int(__import__("os").environ.get("KAN_FUZZ_N", "60"))))
def test_random_geometry_vs_oracle(seed, gpu_lib):
    """Tolerance per tensor = max(stated, 4 x the oracle's own fp32-vs-fp64 difference on that tensor) -- the rule of the golden
    fixtures -- so that draws the reference itself cannot reproduce (InstanceNorm over near-constant or mostly-padding planes)
    are judged against what it can; outputs that sit on a PReLU kink get no upstream gradient."""
    c = _draw(1000 + seed)
    torch.manual_seed(seed)
    kw = dict(groups=c["G"], stride=c["s"], dilation=c["d"], padding=c["p"])
    fam = c["fam"]
    if fam == "KAN":
        kw["base_activation"] = [nn.SiLU, nn.GELU, None][seed % 3]
    layer = K.CONV_KAN_FACTORY[fam](c["C"], c["O"], c["k"], **kw)
    cfg = _cfg(KIND[fam], c["C"], c["O"], k=c["k"], s=c["s"], p=c["p"], d=layer.dilation, groups=c["G"], degree=3,
               extra={"a": 1.0, "b": 1.0} if fam == "JacobiKAN" else {})
    cfg["act"] = ["silu", "gelu", "none"][seed % 3] if fam == "KAN" else "silu" if fam == "FastKAN" else "gelu"
    x = torch.randn(c["B"], c["C"], c["H"], c["W"]) * (1.0 + (seed % 3))
    _check(layer, cfg, x, c)


ALL_FAMILIES = {"KAN": "bspline", "FastKAN": "rbf", "ChebyKAN": "cheby", "BesselKAN": "bessel", "FibonacciKAN": "fibonacci",
                "GegenbauerKAN": "gegenbauer", "HermiteKAN": "hermite", "JacobiKAN": "jacobi", "LaguerreKAN": "laguerre",
                "LucasKAN": "lucas", "TaylorKAN": "taylor", "FourierKAN": "fourier", "LegendreKAN": "legendre",
                "BersnsteinKAN": "bersnstein", "ReLUKAN": "relu", "GRAMKAN": "gram", "WavKAN": "wav"}
EXTRA = {"gegenbauer": {"alpha_param": 0.0}, "laguerre": {"alpha": 1.0}, "jacobi": {"a": 1.0, "b": 1.0},
         "wav": {"wavelet_type": "mexican_hat", "wav_version": "fast"}}


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("KAN_FUZZ_FAM_N", "51"))))
def test_random_family_vs_oracle(seed, gpu_lib):
    """Every registered conv-KAN family (all 17 factory keys besides plain `conv`) on 3x3 'same' layers over the plane sizes and widths of the model zoo (incl. the
    halo / 256-output-tile / position-major paths), affine or plain InstanceNorm."""
    r = random.Random(5000 + seed)
    fam = list(ALL_FAMILIES)[seed % len(ALL_FAMILIES)]
    kind = ALL_FAMILIES[fam]
    G = r.choice([1, 1, 1, 2])
    C = r.choice([2, 4, 6, 16]) * G
    O = r.choice([3, 8, 64, 128, 256]) * G
    H = r.choice([2, 4, 8, 16, 32, 7])
    B = r.choice([1, 3, 16, 33]) if O * H * H <= 65536 else r.choice([1, 2, 4])
    affine = r.random() < 0.4
    torch.manual_seed(seed)
    kw = dict(groups=G, affine=affine)
    layer = K.CONV_KAN_FACTORY[fam](C, O, 3, **kw)
    cfg = _cfg(kind, C, O, k=3, s=1, p=1, d=1, groups=G, degree=3, extra=EXTRA.get(kind, {}),
               act="silu" if kind in ("rbf", "bersnstein") else "gelu")
    if fam == "ReLUKAN":                                          # phases drift apart per channel and plane, as after training
        with torch.no_grad():
            layer.phase_low.add_(0.06 * torch.randn_like(layer.phase_low)); layer.phase_high.add_(0.06 * torch.randn_like(layer.phase_high))
    if fam == "GRAMKAN":                                          # recurrence coefficients away from their ~1e-3 init
        with torch.no_grad():
            layer.beta_weights.normal_(0.0, 0.2)
    if affine:
        with torch.no_grad():
            for m in layer.layer_norm:
                m.weight.add_(0.2 * torch.randn_like(m.weight)); m.bias.add_(0.2 * torch.randn_like(m.bias))
    x = torch.randn(B, C, H, H) * (1.0 + (seed % 2))
    _check(layer, cfg, x, dict(fam=fam, C=C, O=O, G=G, H=H, W=H, B=B, affine=affine))


ONE_D = {"bspline": "KANConv1DLayer", "rbf": "FastKANConv1DLayer", "cheby": "ChebyKANConv1DLayer", "relu": "ReLUKANConv1DLayer"}


@pytest.mark.parametrize("seed", range(_OFF, _OFF + int(__import__("os").environ.get("KAN_FUZZ_1D_N", "24"))))
def test_random_1d_vs_oracle(seed, gpu_lib):
    """The 1-D shims ([B, C, L] lifted to [B, C, 1, L] on the 2-D kernels: kan_layers.py:287-297 and siblings) over random
    lengths, kernels, strides, dilations and groups."""
    r = random.Random(9000 + seed)
    kind = list(ONE_D)[seed % len(ONE_D)]
    G = r.choice([1, 1, 2, 3])
    C, O = r.choice([1, 2, 3, 8]) * G, r.choice([1, 4, 16, 64, 128]) * G
    k = r.choice([1, 3, 3, 5, 7])
    s_, d = r.choice([1, 1, 2, 3]), r.choice([1, 1, 2])
    p = r.choice([0, 1, d * (k - 1) // 2, d * (k - 1)])
    Lx = r.choice([4, 9, 16, 33, 64, 200, 1000])
    if (Lx + 2 * p - d * (k - 1) - 1) // s_ + 1 < 4:                    # keep >= 4 outputs (InstanceNorm over 1-3 values: see above)
        p, s_ = d * (k - 1), 1
    B = r.choice([1, 2, 5, 16]) if O * Lx <= 65536 else r.choice([1, 2])
    torch.manual_seed(seed)
    kw = dict(groups=G, stride=s_, dilation=d, padding=p)
    layer = getattr(K, ONE_D[kind])(C, O, k, **kw)
    cfg = _cfg(kind, C, O, k=k, s=s_, p=p, d=d, groups=G, degree=3, ndim=1, act="gelu" if kind == "bspline" else "silu")
    if kind == "relu":
        with torch.no_grad():
            layer.phase_low.add_(0.05 * torch.randn_like(layer.phase_low)); layer.phase_high.add_(0.05 * torch.randn_like(layer.phase_high))
    x = torch.randn(B, C, Lx) * (1.0 + (seed % 2))
    _check(layer, cfg, x, dict(kind=kind, C=C, O=O, G=G, H=1, W=Lx, k=k, s=s_, d=d, p=p, B=B))


THREE_D = {"bspline": "KANConv3DLayer", "rbf": "FastKANConv3DLayer", "cheby": "ChebyKANConv3DLayer",
           # 3-D shims of the recurrence families (one shared _forward3d; Taylor holds `degree` planes, the others degree + 1) and FourierKAN
           "taylor": "TaylorKANConv3DLayer", "lucas": "LucasKANConv3DLayer", "hermite": "HermiteKANConv3DLayer", "fourier": "FourierKANConv3DLayer"}


@pytest.mark.parametrize("seed", range(_OFF, _OFF + int(__import__("os").environ.get("KAN_FUZZ_3D_N", "18"))))
def test_random_3d_vs_oracle(seed, gpu_lib):
    """The 3-D shims (one 2-D launch set per depth tap: kan_layers.py:261-271 and siblings) over random volumes, kernels,
    strides, dilations, paddings and groups."""
    r = random.Random(11000 + seed)
    kind = list(THREE_D)[seed % len(THREE_D)]
    G = r.choice([1, 1, 2])
    C, O = r.choice([1, 2, 3, 6]) * G, r.choice([1, 4, 16, 64]) * G
    k = r.choice([1, 3, 3])
    s_, d = r.choice([1, 1, 2]), r.choice([1, 1, 2])
    p = r.choice([0, 1, 2])
    D, H, W = r.choice([1, 2, 3, 5, 8]), r.choice([3, 4, 6, 8]), r.choice([3, 4, 7, 8])

    def out(n):
        return (n + 2 * p - d * (k - 1) - 1) // s_ + 1
    if min(out(D), out(H), out(W)) <= 0 or out(D) * out(H) * out(W) < 4:
        p, s_ = d * (k - 1), 1                                    # non-empty output with >= 4 values per InstanceNorm volume
        if out(D) * out(H) * out(W) < 4:
            H, W = 4, 4
    B = r.choice([1, 2, 5])
    torch.manual_seed(seed)
    kw3 = dict(groups=G, stride=s_, dilation=d, padding=p)
    degree = 3
    if kind in ("taylor", "lucas", "hermite"):
        degree = r.choice([1, 2, 3, 4])
        kw3["degree"] = degree
    elif kind == "fourier":
        kw3["grid_size"] = degree
    layer = getattr(K, THREE_D[kind])(C, O, k, **kw3)
    cfg = _cfg(kind, C, O, k=k, s=s_, p=p, d=d, groups=G, degree=degree, ndim=3, act="silu" if kind in ("rbf", "cheby") else "gelu")
    x = torch.randn(B, C, D, H, W) * (1.0 + (seed % 2))
    _check(layer, cfg, x, dict(kind=kind, C=C, O=O, G=G, D=D, H=H, W=W, k=k, s=s_, d=d, p=p, B=B))


@pytest.mark.parametrize("H,G,B", [(16, 1, 3), (8, 1, 5), (8, 2, 2), (4, 1, 6), (32, 1, 1)])
def test_relukan_halo_kernels_vs_oracle(H, G, B, gpu_lib):
    """ReLU-KAN defaults (g = 5, k = 3, SiLU) on the shapes that take the halo forward / halo weight-gradient kernels (compile-time
    spec 9: per-channel phases read from device memory inside the halo expansion), forward, input gradient, weight gradients and
    the phase gradients (weight-gradient kernel on the phase-derivative planes), against the oracle."""
    torch.manual_seed(100 + H + G)
    C, O = 6 * G, 128 * G
    layer = K.CONV_KAN_FACTORY["ReLUKAN"](C, O, 3, groups=G, base_activation=torch.nn.SiLU)     # SiLU: what VGGKAN hands the factory (kan_vgg.py:316)
    from convkan_amd import ops
    geom, basis, plan = ops._plan_cached(layer.conv_spec(), B, C // G, H, H, O // G, C, O)
    assert plan.fwd_halo == (1 if C // G % 2 == 0 else 0) and plan.bwd_weight_halo == (1 if H in (4, 8, 16) else 0), "shape does not reach the halo kernels"
    with torch.no_grad():
        layer.phase_low.add_(0.06 * torch.randn_like(layer.phase_low)); layer.phase_high.add_(0.06 * torch.randn_like(layer.phase_high))
    cfg = _cfg("relu", C, O, k=3, s=1, p=1, d=1, groups=G, degree=3, extra={}, act="silu")
    _check(layer, cfg, torch.randn(B, C, H, H) * 1.5, dict(fam="ReLUKAN", C=C, O=O, G=G, H=H, W=H, B=B, affine=False))


@pytest.mark.parametrize("H,G,B", [(16, 1, 3), (8, 2, 2), (4, 1, 6), (32, 1, 1)])
def test_gramkan_halo_kernels_vs_oracle(H, G, B, gpu_lib):
    """GRAM-KAN defaults (degree 3, SiLU) on the halo-kernel shapes (compile-time spec 10: recurrence coefficients read from device
    memory inside the halo expansion), including the beta_weights gradient (weight-gradient kernel on the coefficient-derivative planes)."""
    torch.manual_seed(200 + H + G)
    C, O = 6 * G, 128 * G
    layer = K.CONV_KAN_FACTORY["GRAMKAN"](C, O, 3, groups=G)
    from convkan_amd import ops
    geom, basis, plan = ops._plan_cached(layer.conv_spec(), B, C // G, H, H, O // G, C, O)
    assert plan.fwd_halo == 1 and plan.bwd_weight_halo == (1 if H in (4, 8, 16) else 0), "shape does not reach the halo kernels"
    with torch.no_grad():
        layer.beta_weights.normal_(0.0, 0.2)
    cfg = _cfg("gram", C, O, k=3, s=1, p=1, d=1, groups=G, degree=3, extra={}, act="silu")
    _check(layer, cfg, torch.randn(B, C, H, H) * 1.5, dict(fam="GRAMKAN", C=C, O=O, G=G, H=H, W=H, B=B, affine=False))


@pytest.mark.parametrize("fam,H,B", [("ReLUKAN", 4, 16), ("ReLUKAN", 2, 128), ("GRAMKAN", 4, 32), ("GRAMKAN", 2, 128)])
def test_parametric_families_on_expanded_position_major_kernels(fam, H, B, gpu_lib):
    """ReLU-KAN / GRAM-KAN defaults on small padded planes: weight gradient (and its phase / coefficient-derivative passes) on the
    expanded position-major operand (k_expand_pm with basis->order = mode), ReLU-KAN's 2x2 forward on it too."""
    from convkan_amd import ops
    torch.manual_seed(300 + H + B)
    C = O = 128
    kw = dict(base_activation=torch.nn.SiLU) if fam == "ReLUKAN" else {}
    layer = K.CONV_KAN_FACTORY[fam](C, O, 3, **kw)
    geom, basis, plan = ops._plan_cached(layer.conv_spec(), B, C, H, H, O, C, O)
    assert plan.bwd_weight_expanded == 1
    assert plan.fwd_expanded == (1 if (fam == "ReLUKAN" and H == 2 and B % 128 == 0) else 0)
    with torch.no_grad():
        if fam == "ReLUKAN":
            layer.phase_low.add_(0.06 * torch.randn_like(layer.phase_low)); layer.phase_high.add_(0.06 * torch.randn_like(layer.phase_high))
        else:
            layer.beta_weights.normal_(0.0, 0.2)
    cfg = _cfg("relu" if fam == "ReLUKAN" else "gram", C, O, k=3, s=1, p=1, d=1, groups=1, degree=3, extra={}, act="silu")
    _check(layer, cfg, torch.randn(B, C, H, H) * 1.5, dict(fam=fam, C=C, O=O, G=1, H=H, W=H, B=B, affine=False))


@pytest.mark.parametrize("seed", range(_OFF, _OFF + int(__import__("os").environ.get("KAN_FUZZ_WAV_N", "40"))))
def test_random_wavkan_vs_oracle(seed, gpu_lib):
    """Wav-KAN over random geometries, the five wavelets and the three weight layouts, scale / translation off their 1 / 0 init."""
    r = random.Random(12000 + seed)
    wavelet = ["mexican_hat", "morlet", "dog", "meyer", "shannon"][seed % 5]
    version = ["fast", "base", "fast_plus_one"][(seed // 5) % 3]
    G = r.choice([1, 1, 2, 3])
    C, O = r.choice([1, 2, 3, 5, 16]) * G, r.choice([1, 3, 4, 6, 33]) * G
    kk = r.choice([1, 3, 3, 5, (3, 1), (1, 3)])
    kh, kw = (kk, kk) if isinstance(kk, int) else kk
    s_, d, p = r.choice([1, 1, 2]), r.choice([1, 1, 2]), r.choice([0, 1, 2])
    H, W = r.choice([(4, 4), (8, 8), (16, 16), (5, 7), (9, 4), (32, 32), (20, 12)])
    if (H + 2 * p - d * (kh - 1) - 1) // s_ + 1 <= 0 or (W + 2 * p - d * (kw - 1) - 1) // s_ + 1 <= 0:
        p = d * (max(kh, kw) - 1)
    ho, wo = (H + 2 * p - d * (kh - 1) - 1) // s_ + 1, (W + 2 * p - d * (kw - 1) - 1) // s_ + 1
    if ho * wo < 4:
        s_, d, p = 1, 1, max(p, max(kh, kw) // 2)
    B = r.choice([1, 2, 3, 9, 20])
    affine = r.random() < 0.4
    torch.manual_seed(seed)
    layer = K.WavKANConv2DLayer(C, O, kk, groups=G, stride=s_, dilation=d, padding=p, wavelet_type=wavelet, wav_version=version,
                                norm_layer=nn.InstanceNorm2d, affine=affine)
    with torch.no_grad():
        for m in layer.wavelet_conv:
            m.scale.add_(0.3 * (torch.rand_like(m.scale) - 0.5)); m.translation.add_(0.5 * torch.randn_like(m.translation))
        if affine:
            for m in layer.layer_norm:
                m.weight.add_(0.2 * torch.randn_like(m.weight)); m.bias.add_(0.2 * torch.randn_like(m.bias))
    cfg = _cfg("wav", C, O, k=kk, s=s_, p=p, d=d, groups=G, extra=dict(wavelet_type=wavelet, wav_version=version))
    x = torch.randn(B, C, H, W) * (0.7 + (seed % 3))
    _check(layer, cfg, x, dict(fam="WavKAN", wavelet=wavelet, version=version, C=C, O=O, G=G, k=kk, s=s_, d=d, p=p, H=H, W=W, B=B, affine=affine))
