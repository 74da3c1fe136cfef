"""CPU oracle for the KAN-conv hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product path (``convkan_amd``) never calls into it and has no
CPU fallback.

What it is: a plain PyTorch-CPU restatement, in this repo's own words, of the op sequence
the reference executes for its three conv-KAN layers.  The convolutions, InstanceNorm and
PReLU themselves live in PyTorch/ATen (an un-vendored dependency of the reference; this
image ships torch 2.10.0), exactly as they do for the reference, so the oracle calls the
same ATen CPU operators.

Parity pin: every function below is checked against the *imported reference itself*
(``/root/reference``, importable in the build container) by
``tests/golden/make_golden.py`` -- which also freezes the reference's outputs as
``tests/golden/*.npz`` -- and against those frozen vectors by ``tests/test_oracle.py``.

Reference sites restated (all paths relative to /root/reference):
  * layers/kan_layers.py:184-190   knot vector               -> bspline_knots
  * layers/kan_layers.py:203-236   order-0 indicator + Cox-de Boor -> bspline_basis
  * layers/kan_layers.py:197-247   forward_kan               -> kan_conv2d
  * layers/kan_layers.py:249-258   group split / concat      -> _per_group
  * utils/utils.py:19-33           RadialBasisFunction       -> rbf_grid / rbf_basis
  * layers/fast_kan_layers.py:100-120 forward_fast_kan       -> fastkan_conv2d
  * layers/cheby_kan_layers.py:91-111 forward_ChebyKAN       -> chebykan_conv2d
"""
from __future__ import annotations

import math
from typing import Callable, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- bases
def bspline_knots(grid_size: int, spline_order: int, grid_range: Sequence[float]) -> Tensor:
    """Uniform extended knot vector (kan_layers.py:184-190): G + 2S + 1 fp32 knots."""
    lo, hi = float(grid_range[0]), float(grid_range[1])
    step = (hi - lo) / grid_size
    return torch.linspace(lo - step * spline_order, hi + step * spline_order,
                          grid_size + 2 * spline_order + 1, dtype=torch.float32)


def bspline_basis(x: Tensor, knots: Tensor, spline_order: int) -> Tensor:
    """B-spline bases of ``x`` appended as a last axis of length G+S (kan_layers.py:203-236).

    Level 0 is the half-open interval indicator; level k blends neighbours with the
    Cox-de Boor weights.  Zero-width knot spans get denominator 1 as in the reference.
    """
    g = knots.to(device=x.device, dtype=x.dtype)
    xe = x.unsqueeze(-1)
    basis = torch.logical_and(xe >= g[:-1], xe < g[1:]).to(x.dtype)
    one = torch.ones((), dtype=x.dtype, device=x.device)
    for k in range(1, spline_order + 1):
        left, right = g[:-(k + 1)], g[k + 1:]
        span_l = g[k:-1] - left
        span_r = right - g[1:-k]
        span_l = torch.where(span_l == 0, one, span_l)
        span_r = torch.where(span_r == 0, one, span_r)
        basis = (xe - left) / span_l * basis[..., :-1] + (right - xe) / span_r * basis[..., 1:]
    return basis


def rbf_grid(grid_size: int, grid_range: Sequence[float]):
    """Centres and width of the FastKAN RBFs (utils/utils.py:28-30)."""
    lo, hi = float(grid_range[0]), float(grid_range[1])
    centres = torch.linspace(lo, hi, grid_size)
    denom = (hi - lo) / (grid_size - 1)
    return centres, denom


def rbf_basis(x: Tensor, centres: Tensor, denom: float) -> Tensor:
    """exp(-((x - c_g)/d)^2) on a new last axis (utils/utils.py:32-33)."""
    return torch.exp(-(((x.unsqueeze(-1) - centres.to(x)) / denom) ** 2))


def cheby_basis(x: Tensor, degree: int, eps: float = 1e-7) -> Tensor:
    """T_k(tanh x) = cos(k acos(clamp(tanh x))) on a new axis 2 (cheby_kan_layers.py:93-96)."""
    theta = torch.acos(torch.clamp(torch.tanh(x).unsqueeze(2), -1 + eps, 1 - eps))
    k = torch.arange(0, degree + 1, device=x.device).view(1, 1, -1, *([1] * (x.dim() - 2)))
    return torch.cos(theta * k)


def _planes_to_channels(b: Tensor) -> Tensor:
    """[B,C,H,W,K] -> [B,C*K,H,W] with channel index c*K+k (kan_layers.py:237)."""
    return b.movedim(-1, 2).flatten(1, 2)


# --------------------------------------------------------------------------- layers
def _per_group(x: Tensor, groups: int, fn: Callable[[Tensor, int], Tensor]) -> Tensor:
    parts = torch.chunk(x, groups, dim=1) if groups > 1 else (x,)
    return torch.cat([fn(p, g) for g, p in enumerate(parts)], dim=1)


def _conv(x, w, stride, padding, dilation):
    """conv2d; conv3d for the 3-D shims (kan_layers.py:261-271: the same layer with nn.Conv3d / InstanceNorm3d)."""
    return (F.conv3d if w.dim() == 5 else F.conv2d)(x, w, None, stride, padding, dilation, 1)


def kan_conv2d(x: Tensor, w_base: Sequence[Tensor], w_spline: Sequence[Tensor], prelu_a: Sequence[Tensor],
               *, knots: Tensor, spline_order: int, act: Optional[Callable[[Tensor], Tensor]],
               stride=1, padding=0, dilation=1, groups: int = 1,
               norm: Optional[Sequence[Callable[[Tensor], Tensor]]] = None,
               pre_norm_out: Optional[list] = None, prelu_gate: Optional[Tensor] = None) -> Tensor:
    """B-spline KAN conv layer (kan_layers.py:197-258).

    ``norm[g]`` is the output normalisation of group g; ``None`` means InstanceNorm2d with
    eps 1e-5, no affine (the reference's constructor default).  ``pre_norm_out`` (a list)
    receives the pre-normalisation sums, for tests of the conv stage alone.
    ``prelu_gate`` (bool, shaped like the output; single-group layers): take the PReLU branch the caller names
    instead of testing n > 0 -- parity tests use it to differentiate the SAME piecewise-linear branch as the
    implementation under test where a normalised value sits within rounding noise of the kink (the value changes
    by (1 - a)|n| ~ 1e-7 there, the gradient by O(1)).
    """
    def one(xg, g):
        a = xg if act is None else act(xg)
        z = _conv(a, w_base[g], stride, padding, dilation)
        planes = _planes_to_channels(bspline_basis(xg, knots, spline_order).contiguous())
        z = z + _conv(planes, w_spline[g], stride, padding, dilation)
        if pre_norm_out is not None:
            pre_norm_out.append(z)
        n = F.instance_norm(z, eps=1e-5) if norm is None else norm[g](z)
        if prelu_gate is not None:
            assert groups == 1
            return torch.where(prelu_gate, n, prelu_a[g] * n)
        return F.prelu(n, prelu_a[g])
    return _per_group(x, groups, one)


def fastkan_conv2d(x: Tensor, w_base: Sequence[Tensor], w_spline: Sequence[Tensor],
                   *, centres: Tensor, denom: float, act: Optional[Callable[[Tensor], Tensor]],
                   stride=1, padding=0, dilation=1, groups: int = 1,
                   norm: Optional[Sequence[Callable[[Tensor], Tensor]]] = None) -> Tensor:
    """FastKAN conv layer (fast_kan_layers.py:100-120): norm acts on the *input* of the RBFs."""
    def one(xg, g):
        a = xg if act is None else act(xg)
        z = _conv(a, w_base[g], stride, padding, dilation)
        xn = F.instance_norm(xg, eps=1e-5) if norm is None else norm[g](xg)
        planes = _planes_to_channels(rbf_basis(xn, centres, denom))
        return z + _conv(planes, w_spline[g], stride, padding, dilation)
    return _per_group(x, groups, one)


def chebykan_conv2d(x: Tensor, w_poly: Sequence[Tensor], *, degree: int,
                    stride=1, padding=0, dilation=1, groups: int = 1,
                    norm: Optional[Sequence[Callable[[Tensor], Tensor]]] = None,
                    pre_norm_out: Optional[list] = None) -> Tensor:
    """ChebyKAN conv layer (cheby_kan_layers.py:91-111): no base branch, no activation."""
    def one(xg, g):
        planes = cheby_basis(xg, degree).flatten(1, 2)
        z = _conv(planes, w_poly[g], stride, padding, dilation)
        if pre_norm_out is not None:
            pre_norm_out.append(z)
        return F.instance_norm(z, eps=1e-5) if norm is None else norm[g](z)
    return _per_group(x, groups, one)


# --------------------------------------------------------------------------- polynomial families (section 8(f), rank 3)
def poly_basis(x: Tensor, family: str, degree: int, **kw) -> Tensor:
    """Basis planes [B, C, n, H, W] of the three-term-recurrence conv-KAN families, each restated from its own file
    (all evaluate on t = tanh(x)):
      bessel      bessel_kan_layers.py compute_bessel_basis:        y0 = 1, y1 = t + 1, yn = (2n-1) t y(n-1) + y(n-2)
      fibonacci   fibonacci_kan_layers.py compute_fibonacci_basis:  F0 = 0, F1 = 1, Fn = t F(n-1) + F(n-2)
      gegenbauer  gegenbauer_kan_layers.py compute_gegenbauer_basis: C0 = 1, C1 = 2 a t, C(n+1) = (2(n+a) t Cn - (n+2a-1) C(n-1))/(n+1)
      hermite     hermite_kan_layers.py:117-146:                     H0 = 1, H1 = 2t, Hn = 2t H(n-1) - 2(n-1) H(n-2)
      laguerre    laguerre_kan_layers.py compute_laguerre_basis:     L0 = 1, L1 = 1 + a - t, Lk = ((2k-1+a-t) L(k-1) - (k-1+a) L(k-2))/k
      lucas       lucas_kan_layers.py:140-174:                       L0 = 2, L1 = t, Ln = t L(n-1) + L(n-2)
      taylor      taylor_kan_layers.py compute_taylor_basis:         `degree` planes t^0 .. t^(degree-1)
      jacobi      jacobi_kan_layers.py:118-136:                      P0 = 1, P1 = ((a-b) + (a+b+2) t)/2, Pi = (th t + th1) P(i-1) - th2 P(i-2)
    """
    t = torch.tanh(x)
    one = torch.ones_like(t)
    if family == "taylor":
        planes = [one]
        for _ in range(1, degree):
            planes.append(planes[-1] * t if len(planes) > 1 else t)
        return torch.stack(planes[:degree], dim=2)
    if family == "bessel":
        p = [one, t + 1]
        for i in range(2, degree + 1):
            p.append((2 * i - 1) * t * p[i - 1] + p[i - 2])
    elif family == "fibonacci":
        p = [torch.zeros_like(t), one]
        for i in range(2, degree + 1):
            p.append(t * p[i - 1] + p[i - 2])
    elif family == "gegenbauer":
        a = kw["alpha_param"]
        p = [one, 2 * a * t]
        for n in range(1, degree):
            p.append((2 * (n + a) * t * p[n] - (n + 2 * a - 1) * p[n - 1]) / (n + 1))
    elif family == "hermite":
        p = [one, 2 * t]
        for i in range(2, degree + 1):
            p.append(2 * t * p[i - 1] - 2 * (i - 1) * p[i - 2])
    elif family == "laguerre":
        a = kw["alpha"]
        p = [one, (1 + a) - t]
        for k in range(2, degree + 1):
            p.append(((2 * (k - 1) + 1 + a - t) * p[k - 1] - (k - 1 + a) * p[k - 2]) / k)
    elif family == "lucas":
        p = [2 * one, t]
        for i in range(2, degree + 1):
            p.append(t * p[i - 1] + p[i - 2])
    elif family == "jacobi":
        a, b = kw["a"], kw["b"]
        p = [one, ((a - b) + (a + b + 2) * t) / 2]
        for i in range(2, degree + 1):
            th = (2 * i + a + b) * (2 * i + a + b - 1) / (2 * i * (i + a + b))
            th1 = (2 * i + a + b - 1) * (a * a - b * b) / (2 * i * (i + a + b) * (2 * i + a + b - 2))
            th2 = (i + a - 1) * (i + b - 1) * (2 * i + a + b) / (i * (i + a + b) * (2 * i + a + b - 2))
            p.append((th * t + th1) * p[i - 1] - th2 * p[i - 2])
    else:
        raise ValueError(family)
    return torch.stack(p[:degree + 1], dim=2)


def polykan_conv2d(x: Tensor, w_base: Sequence[Tensor], w_poly: Sequence[Tensor], prelu_a: Sequence[Tensor], *, family: str,
                   degree: int, act: Optional[Callable[[Tensor], Tensor]], stride=1, padding=0, dilation=1, groups: int = 1,
                   norm: Optional[Sequence[Callable[[Tensor], Tensor]]] = None, pre_norm_out: Optional[list] = None, **kw) -> Tensor:
    """PReLU(norm(conv(act(x), W_b) + conv(basis(tanh x), W_p))), poly channel = c*n + k  (e.g. lucas_kan_layers.py:176-193)."""
    def one(xg, g):
        a = xg if act is None else act(xg)
        z = _conv(a, w_base[g], stride, padding, dilation) + _conv(poly_basis(xg, family, degree, **kw).flatten(1, 2), w_poly[g],
                                                                   stride, padding, dilation)
        if pre_norm_out is not None:
            pre_norm_out.append(z)
        n = F.instance_norm(z, eps=1e-5) if norm is None else norm[g](z)
        return F.prelu(n, prelu_a[g])
    return _per_group(x, groups, one)


def jacobikan_conv2d(x: Tensor, w_base: Sequence[Tensor], poly_weights: Tensor, *, degree: int, a: float, b: float,
                     act: Optional[Callable[[Tensor], Tensor]], stride=1, padding=0, dilation=1, groups: int = 1,
                     norm: Optional[Sequence[Callable[[Tensor], Tensor]]] = None, pre_norm_out: Optional[list] = None) -> Tensor:
    """act(norm(conv(x, W_b) + conv(P(tanh x), poly_weights[g]))), planes concatenated plane-major k*C + c
    (jacobi_kan_layers.py:136-166)."""
    def one(xg, g):
        planes = poly_basis(xg, "jacobi", degree, a=a, b=b).transpose(1, 2).flatten(1, 2)       # [B, n*C, H, W]
        z = _conv(xg, w_base[g], stride, padding, dilation) + _conv(planes, poly_weights[g], stride, padding, dilation)
        if pre_norm_out is not None:
            pre_norm_out.append(z)
        n = F.instance_norm(z, eps=1e-5) if norm is None else norm[g](z)
        return n if act is None else act(n)
    return _per_group(x, groups, one)


def fourier_basis(x: Tensor, grid_size: int) -> Tensor:
    """[B, C, 2G, H, W]: cos(k x), k = 1..G, then sin(k x) (fourier_kan_layers.py:163-187)."""
    k = torch.arange(1, grid_size + 1, device=x.device, dtype=x.dtype).view(1, 1, grid_size, *([1] * (x.dim() - 2)))
    kx = k * x.unsqueeze(2)
    return torch.cat((torch.cos(kx), torch.sin(kx)), dim=2)


def fourierkan_conv2d(x: Tensor, w_base: Sequence[Tensor], w_fourier: Sequence[Tensor], prelu_a: Sequence[Tensor], *, grid_size: int,
                      act: Optional[Callable[[Tensor], Tensor]], stride=1, padding=0, dilation=1, groups: int = 1,
                      norm: Optional[Sequence[Callable[[Tensor], Tensor]]] = None, pre_norm_out: Optional[list] = None) -> Tensor:
    """PReLU(norm(conv(act(x), W_b) + conv(fourier(x), W_f)))  (fourier_kan_layers.py:189-205)."""
    def one(xg, g):
        a = xg if act is None else act(xg)
        z = _conv(a, w_base[g], stride, padding, dilation) + _conv(fourier_basis(xg, grid_size).flatten(1, 2), w_fourier[g],
                                                                   stride, padding, dilation)
        if pre_norm_out is not None:
            pre_norm_out.append(z)
        n = F.instance_norm(z, eps=1e-5) if norm is None else norm[g](z)
        return F.prelu(n, prelu_a[g])
    return _per_group(x, groups, one)


def relukan_basis(x: Tensor, phase_low: Tensor, phase_high: Tensor, r: float) -> Tensor:
    """[B, C, g+k, H, W]: (relu(x - lo) * relu(hi - x) * r)^2, phases of shape (1, C, g+k, 1, 1)  (relu_kan_layers.py:126-131)."""
    xe = x.unsqueeze(2)
    q = torch.relu(xe - phase_low) * torch.relu(phase_high - xe) * r
    return q * q


def relukan_conv2d(x: Tensor, w_base: Sequence[Tensor], w_relukan: Sequence[Tensor], phase_low: Tensor, phase_high: Tensor, *,
                   g: int, k: int, act: Optional[Callable[[Tensor], Tensor]], stride=1, padding=0, dilation=1, groups: int = 1,
                   norm: Optional[Sequence[Callable[[Tensor], Tensor]]] = None, pre_norm_out: Optional[list] = None) -> Tensor:
    """act(norm(conv(act(x), W_b) + conv(relukan_basis(x), W_r))), one phase pair shared by all groups
    (relu_kan_layers.py:118-146)."""
    r = 4 * g * g / ((k + 1) * (k + 1))

    def one(xg, gi):
        a = xg if act is None else act(xg)
        z = _conv(a, w_base[gi], stride, padding, dilation) + _conv(relukan_basis(xg, phase_low, phase_high, r).flatten(1, 2),
                                                                    w_relukan[gi], stride, padding, dilation)
        if pre_norm_out is not None:
            pre_norm_out.append(z)
        n = F.instance_norm(z, eps=1e-5) if norm is None else norm[gi](z)
        return n if act is None else act(n)
    return _per_group(x, groups, one)


def gram_basis(t: Tensor, beta_weights: Tensor, degree: int) -> Tensor:
    """[B, C, degree+1, H, W]: Gram polynomials of t, p_i = t p_{i-1} - beta(i-1, i) p_{i-2},
    beta(n, m) = (m+n)(m-n) n^2 / (m^2 / (4 n^2 - 1)) * beta_weights[n]  (gram_kan_layers.py:150-170)."""
    p0, p1 = torch.ones_like(t), t
    out = [p0, p1]
    for i in range(2, degree + 1):
        n, m = i - 1, i
        beta = (((m + n) * (m - n) * n ** 2) / (m ** 2 / (4.0 * n ** 2 - 1.0))) * beta_weights[n]
        p2 = t * p1 - beta * p0
        out.append(p2)
        p0, p1 = p1, p2
    return torch.stack(out[:degree + 1], dim=2)


def gramkan_conv2d(x: Tensor, w_base: Sequence[Tensor], poly_weights: Tensor, beta_weights: Tensor, *, degree: int,
                   act: Optional[Callable[[Tensor], Tensor]], stride=1, padding=0, dilation=1, groups: int = 1,
                   norm: Optional[Sequence[Callable[[Tensor], Tensor]]] = None, pre_norm_out: Optional[list] = None) -> Tensor:
    """act(norm(conv(act(x), W_b) + conv(act(P(tanh x)), poly_weights[g]))), planes concatenated plane-major k*C + c
    (gram_kan_layers.py:172-189)."""
    f = (lambda v: v) if act is None else act

    def one(xg, g):
        planes = f(gram_basis(torch.tanh(xg), beta_weights, degree)).transpose(1, 2).flatten(1, 2)       # [B, n*C, H, W]
        z = _conv(f(xg), w_base[g], stride, padding, dilation) + _conv(planes, poly_weights[g], stride, padding, dilation)
        if pre_norm_out is not None:
            pre_norm_out.append(z)
        n = F.instance_norm(z, eps=1e-5) if norm is None else norm[g](z)
        return f(n)
    return _per_group(x, groups, one)


def legendrekan_conv2d(x: Tensor, w_base: Sequence[Tensor], poly_weights: Tensor, *, degree: int, stride=1, padding=0, dilation=1,
                       groups: int = 1, norm: Optional[Sequence[Callable[[Tensor], Tensor]]] = None,
                       pre_norm_out: Optional[list] = None) -> Tensor:
    """legendre_kan_layers.py:126-158: SiLU(norm(conv(x, W_b) + conv(P(x_n), poly_weights[g]))), x_n = 2(x - min)/(max - min) - 1
    over the whole group tensor (:130), P_0 = 1, P_1 = x_n, P_{n+1} = ((2n+1) x_n P_n - n P_{n-1})/(n+1) (:109-124), planes
    concatenated plane-major."""
    def one(xg, g):
        xn = 2 * (xg - xg.min()) / (xg.max() - xg.min()) - 1
        p = [torch.ones_like(xn), xn]
        for n in range(1, degree):
            p.append(((2.0 * n + 1.0) * xn * p[-1] - n * p[-2]) / (n + 1.0))
        z = _conv(xg, w_base[g], stride, padding, dilation) + _conv(torch.cat(p, dim=1), poly_weights[g], stride, padding, dilation)
        if pre_norm_out is not None:
            pre_norm_out.append(z)
        return F.silu(F.instance_norm(z, eps=1e-5) if norm is None else norm[g](z))
    return _per_group(x, groups, one)


def bersnsteinkan_conv2d(x: Tensor, w_base: Sequence[Tensor], poly_weights: Tensor, *, degree: int,
                         act: Optional[Callable[[Tensor], Tensor]], stride=1, padding=0, dilation=1, groups: int = 1,
                         norm: Optional[Sequence[Callable[[Tensor], Tensor]]] = None, pre_norm_out: Optional[list] = None) -> Tensor:
    """bersnstein_kan_layers.py:120-169, restated literally: the de Casteljau sweep over all-ones start coefficients on
    t = sigmoid(x) (which leaves every plane at 1 up to rounding), planes channel-major c*(degree+1) + k."""
    def one(xg, g):
        t = torch.sigmoid(xg).unsqueeze(-1)
        b = torch.ones(xg.shape + (degree + 1,), dtype=xg.dtype, device=xg.device)
        for j in range(1, degree + 1):
            m = degree + 1 - j
            b = torch.cat((b[..., :m] * (1 - t) + b[..., 1:m + 1] * t, b[..., m:]), dim=-1)
        planes = b.moveaxis(-1, 2).flatten(1, 2)
        z = _conv(xg, w_base[g], stride, padding, dilation) + _conv(planes, poly_weights[g], stride, padding, dilation)
        if pre_norm_out is not None:
            pre_norm_out.append(z)
        n = F.instance_norm(z, eps=1e-5) if norm is None else norm[g](z)
        return n if act is None else act(n)
    return _per_group(x, groups, one)


# --------------------------------------------------------------------------- whole-model oracle
def wavelet(u: Tensor, kind: str, n_channels: int = 1) -> Tensor:
    """The five wavelets of wav_kan_layers.py:145-190 on u = (x - translation) / scale, u: [B, O, C, H, W]."""
    if kind == "mexican_hat":
        return (2 / (math.sqrt(3) * math.pi ** 0.25)) * ((u ** 2) - 1) * torch.exp(-0.5 * u ** 2)
    if kind == "morlet":
        return torch.exp(-0.5 * u ** 2) * torch.cos(5.0 * u)
    if kind == "dog":
        return -u * torch.exp(-0.5 * u ** 2)
    if kind == "meyer":
        v = torch.abs(u)
        nu = lambda t: t ** 4 * (35 - 84 * t + 70 * t ** 2 - 20 * t ** 3)
        aux = torch.where(v <= 1 / 2, torch.ones_like(v), torch.where(v >= 1, torch.zeros_like(v), torch.cos(math.pi / 2 * nu(2 * v - 1))))
        return torch.sin(math.pi * v) * aux
    if kind == "shannon":                              # the Hamming window runs over the CHANNEL axis (x.size(2), wav_kan_layers.py:180-186)
        window = torch.hamming_window(u.size(2), periodic=False, dtype=u.dtype, device=u.device).view(1, 1, -1, *([1] * (u.dim() - 3)))
        return torch.sinc(u / math.pi) * window
    raise ValueError(kind)


def wavkan_conv2d(x: Tensor, w_base: Sequence[Tensor], scale: Sequence[Tensor], translation: Sequence[Tensor], w_wavelet: Sequence[Tensor],
                  w_out: Sequence[Tensor], *, wavelet_type: str, stride=1, padding=0, dilation=1, groups: int = 1,
                  norm: Optional[Sequence[Callable[[Tensor], Tensor]]] = None, pre_norm_out: Optional[list] = None) -> Tensor:
    """norm(wavelet_out(sum_c conv(psi((x_c - t_oc) / s_oc), Wk[o, c])) + conv(SiLU(x), W_b))  (wav_kan_layers.py:430-443; the three
    wavelet-conv versions, :186-217 / :257-276 / :318-338, are this sum over differently shaped weights).  scale / translation: per group
    [1, O, C, 1, 1]; w_wavelet: per group [O, C, kh, kw]; w_out: per group [O, O, 1, 1] -- or their 3-D counterparts ([B, C, D, H, W] input,
    one more unit / kernel axis: WavKANConv3DLayer, :457-464)."""
    def one(xg, g):
        base = _conv(F.silu(xg), w_base[g], stride, padding, dilation)
        psi = wavelet((xg.unsqueeze(1) - translation[g]) / scale[g], wavelet_type)           # [B, O, C, H, W]
        O = psi.shape[1]
        u = torch.cat([_conv(psi[:, o], w_wavelet[g][o:o + 1], stride, padding, dilation) for o in range(O)], dim=1)
        z = _conv(u, w_out[g], 1, 0, 1) + base                # the 1x1(x1) `wavelet_out` conv
        if pre_norm_out is not None:
            pre_norm_out.append(z)
        return F.instance_norm(z, eps=1e-5) if norm is None else norm[g](z)
    return _per_group(x, groups, one)


def wavkan_param_views(sd, groups: int, wav_version: str, ndim: int = 2):
    """(w_base, scale, translation, w_wavelet [O, C, kh, kw], w_out [O, O, 1, 1]) per group from a WavKANConv{1,2}DLayer's named
    parameters `sd` (wav_kan_layers.py:117-139 'base': one Conv(C -> 1) per output; :303-310 'fast': grouped conv [O, C, k, k]; :243-250
    'fast_plus_one': Conv(N+1)D [O, 1, C, k, k]).  1-D layers are lifted to [.., 1, L] views."""
    lift = (lambda t: t.unsqueeze(-2)) if ndim == 1 else (lambda t: t)
    out = ([], [], [], [], [])
    for g in range(groups):
        pre = f"wavelet_conv.{g}."
        if wav_version == "base":
            n_out = sd[pre + "scale"].shape[1]
            wk = torch.cat([sd[pre + f"wavelet_weights.{o}.weight"] for o in range(n_out)], dim=0)
        elif wav_version == "fast":
            wk = sd[pre + "wavelet_weights.weight"]
        else:
            wk = sd[pre + "wavelet_weights.weight"].squeeze(1)
        for lst, t in zip(out, (sd[f"base_conv.{g}.weight"], sd[pre + "scale"], sd[pre + "translation"], wk, sd[pre + "wavelet_out.weight"])):
            lst.append(lift(t))
    return out


VGG11_CFG = [64, "M", 128, "M", 256, 256, "M", 512, 512, "M", 512, 512]


class OracleKANConv2d(torch.nn.Module):
    """Parameter holder + ``kan_conv2d`` call; used by the CPU baseline model only."""

    def __init__(self, cin, cout, kernel_size=3, padding=1, grid_size=5, spline_order=3,
                 grid_range=(-1.0, 1.0), act=F.silu):
        super().__init__()
        nb = grid_size + spline_order
        self.w_base = torch.nn.Parameter(torch.empty(cout, cin, kernel_size, kernel_size))
        self.w_spline = torch.nn.Parameter(torch.empty(cout, cin * nb, kernel_size, kernel_size))
        self.prelu = torch.nn.Parameter(torch.full((1,), 0.25))
        torch.nn.init.kaiming_uniform_(self.w_base, nonlinearity="linear")
        torch.nn.init.kaiming_uniform_(self.w_spline, nonlinearity="linear")
        self.knots = bspline_knots(grid_size, spline_order, grid_range)
        self.spline_order, self.padding, self.act = spline_order, padding, act

    def forward(self, x, prelu_gate=None):
        return kan_conv2d(x, [self.w_base], [self.w_spline], [self.prelu], knots=self.knots,
                          spline_order=self.spline_order, act=self.act, padding=self.padding, prelu_gate=prelu_gate)


class OracleKANVGG(torch.nn.Module):
    """KAN-VGG (models/kan_vgg.py:73-188 composition): KAN convs + MaxPool(2,2), then
    AdaptiveAvgPool(1,1) -> flatten -> Dropout(0.5) -> Linear."""

    def __init__(self, cfg=VGG11_CFG, in_ch=3, num_classes=10, width_scale=1):
        super().__init__()
        feats, c = [], in_ch
        for v in cfg:
            if v == "M":
                feats.append(torch.nn.MaxPool2d(2, 2))
            else:
                feats.append(OracleKANConv2d(c, v * width_scale))
                c = v * width_scale
        self.features = torch.nn.ModuleList(feats)
        self.classifier = torch.nn.Sequential(torch.nn.Dropout(0.5), torch.nn.Linear(c, num_classes))

    def forward(self, x):
        for f in self.features:
            x = f(x)
        x = F.adaptive_avg_pool2d(x, (1, 1)).flatten(1)
        return self.classifier(x)


# ------------------------------------------------------------------------------------ MLP KANLayer (section 8(f), rank 2)
def kan_linear(x: Tensor, base_weight: Tensor, spline_weight: Tensor, ln_weight: Tensor, ln_bias: Tensor, prelu_a: Tensor, *,
               grid_size: int = 5, spline_order: int = 3, grid_range: Sequence[float] = (-1, 1),
               act: Optional[Callable[[Tensor], Tensor]] = F.gelu, eps: float = 1e-5, pre: Optional[list] = None) -> Tensor:
    """layers/kan_layers.py:48-114 (KANLayer.forward): x [B, I]; base_weight [O, I]; spline_weight [O, I, G+S].

    base = act(x) W_b^T (53-54); bases by the same indicator + Cox-de Boor recursion as the conv layer (62-90);
    spline = bases.view(B, I*(G+S)) W_s.view(O, I*(G+S))^T (104-106); LayerNorm(O) then scalar PReLU (108-110)."""
    knots = bspline_knots(grid_size, spline_order, grid_range)
    base = F.linear(act(x) if act is not None else x, base_weight)
    bases = bspline_basis(x, knots, spline_order)                       # [B, I, G+S]
    spline = F.linear(bases.reshape(x.shape[0], -1), spline_weight.reshape(spline_weight.shape[0], -1))
    z = base + spline
    if pre is not None:
        pre.append(z.detach().clone())
    return F.prelu(F.layer_norm(z, (z.shape[1],), ln_weight, ln_bias, eps), prelu_a)


def flops_kan_conv(B, C, O, Ho, Wo, kh, kw, n_planes):
    """Dense algorithmic FLOPs of one conv stage, forward only."""
    return 2.0 * B * O * Ho * Wo * C * n_planes * kh * kw


__all__ = [n for n in dir() if not n.startswith("_")]
_ = math
