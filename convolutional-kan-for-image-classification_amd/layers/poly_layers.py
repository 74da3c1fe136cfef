"""Polynomial-basis conv-KAN layers whose basis is a three-term recurrence -- SURVEY.md section 8(f), "next" rank 3.

  reference class (layers/...)                                this file
  bessel_kan_layers.py:38-190      BesselKANConvNDLayer/2D      BesselKANConvNDLayer / BesselKANConv2DLayer
  fibonacci_kan_layers.py:41-240   FibonacciKANConvNDLayer/2D   FibonacciKANConvNDLayer / FibonacciKANConv2DLayer
  gegenbauer_kan_layers.py:..-225  GegenbauerKANConvNDLayer/2D  GegenbauerKANConvNDLayer / GegenbauerKANConv2DLayer
  hermite_kan_layers.py:30-183     HermiteKANConvNDLayer/2D     HermiteKANConvNDLayer / HermiteKANConv2DLayer
  laguerre_kan_layers.py:38-202    LaguerreKANConvNDLayer/2D    LaguerreKANConvNDLayer / LaguerreKANConv2DLayer
  lucas_kan_layers.py:40-218       LucasKANConvNDLayer/2D       LucasKANConvNDLayer / LucasKANConv2DLayer
  taylor_kan_layers.py:40-195      TaylorKANConvNDLayer/2D      TaylorKANConvNDLayer / TaylorKANConv2DLayer
  jacobi_kan_layers.py:55-199      JacobiKANConvNDLayer/2D      JacobiKANConvNDLayer / JacobiKANConv2DLayer
  fourier_kan_layers.py:63-231     FourierKANConvNDLayer/2D     FourierKANConvNDLayer / FourierKANConv2DLayer  (own basis kind)
  legendre_kan_layers.py:50-183    LegendreKANConvNDLayer/2D    LegendreKANConvNDLayer / LegendreKANConv2DLayer
  bersnstein_kan_layers.py:63-201  BersnsteinKANConvNDLayer/2D  BersnsteinKANConvNDLayer / BersnsteinKANConv2DLayer

The first seven share one shape (e.g. lucas_kan_layers.py:176-199):
    y = Dropout(PReLU(norm(conv(act(x), W_base) + conv(basis(tanh x), W_poly))))
with basis planes T_0..T_degree (Taylor: x^0..x^(degree-1)) of t = tanh(x), channel index c*(degree+1)+k.  Every one of
these bases is T_k = (A_k t + B_k) T_{k-1} + C_k T_{k-2}; the coefficients are computed here and handed to the same HIP
conv stage as the B-spline layers (KAN_BASIS_POLY), so forward, input gradient and weight gradient are the fused kernels.
Jacobi keeps its own structure (identity base branch, plane-major ``poly_weights``, norm then activation).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops
from .conv_layers import (_HipLayer, _act_code, _check_groups, _dropout2d, _filter_norm_kwargs, _fusable_instnorm, _norm3d, _unfused_pool,
                          conv3d_stage)
from .conv_layers import _need_conv2d as _need_conv1d_or_2d


def _need_conv2d(conv_class, ndim):
    if conv_class is not nn.Conv2d or ndim != 2:
        raise NotImplementedError("this polynomial-family layer is built for 2-D only (1-D/3-D: SURVEY.md 8(f))")

Coeffs = Tuple[float, float, float, List[Tuple[float, float, float]]]       # c0, a1, b1, [(A_k, B_k, C_k) for k = 2..]


def _table(c: Coeffs, n_basis: int) -> Tuple[float, ...]:
    c0, a1, b1, rec = c
    out = [float(c0), float(a1), float(b1)]
    for k in range(2, n_basis):
        out.extend(float(v) for v in rec[k - 2])
    return tuple(out)


class _RecurrenceKANConvNDLayer(_HipLayer):
    """Shared body of the seven 'base conv + polynomial conv -> norm -> PReLU' layers."""
    _min_degree = 0
    _min_degree_msg = 'degree must be non-negative'

    def _setup(self, conv_class, norm_class, input_dim, output_dim, kernel_size, degree, groups, padding, stride, dilation, ndim,
               base_activation, dropout, norm_kwargs, none_is_identity=True):
        _need_conv1d_or_2d(conv_class, ndim, allow_3d=True)      # 1-D layers run the 2-D kernels on [B, C, 1, L] (_HipLayer); 3-D: conv3d_stage
        _check_groups(groups, input_dim, output_dim)
        if degree < self._min_degree:
            raise ValueError(self._min_degree_msg)
        self.input_dim, self.output_dim, self.kernel_size, self.degree = input_dim, output_dim, kernel_size, degree
        self.groups, self.padding, self.stride, self.dilation, self.ndim = groups, padding, stride, dilation, ndim
        self.base_activation = base_activation() if (base_activation is not None or not none_is_identity) else nn.Identity()
        self.norm_kwargs = norm_kwargs
        self.input_dim_group, self.output_dim_group = input_dim // groups, output_dim // groups
        self.poly_input_dim_group = self.input_dim_group * self._n_planes()
        cg, og = self.input_dim_group, self.output_dim_group
        self.base_conv = nn.ModuleList([conv_class(cg, og, kernel_size, stride, padding, dilation, groups=1, bias=False)
                                        for _ in range(groups)])
        self.poly_conv = nn.ModuleList([conv_class(self.poly_input_dim_group, og, kernel_size, stride, padding, dilation, groups=1,
                                                   bias=False) for _ in range(groups)])
        self.layer_norm = nn.ModuleList([norm_class(og, **_filter_norm_kwargs(norm_class, norm_kwargs)) for _ in range(groups)])
        self.prelus = nn.ModuleList([nn.PReLU() for _ in range(groups)])
        self.dropout = _dropout2d(dropout, ndim)
        for conv in self.base_conv:
            nn.init.kaiming_uniform_(conv.weight, nonlinearity='linear')
        for conv in self.poly_conv:
            nn.init.kaiming_uniform_(conv.weight, nonlinearity='linear')
        self._act_code = _act_code(self.base_activation, host_ok=True)
        if self._n_planes() > 11:
            raise NotImplementedError("the HIP recurrence basis holds at most 11 planes per channel (degree <= 10)")

    def _n_planes(self) -> int:
        return self.degree + 1

    def _coeffs(self) -> Coeffs:
        raise NotImplementedError

    def conv_spec(self) -> ops.ConvSpec:
        n = self._n_planes()
        return self._spec(kind=L.BASIS_POLY, n_basis=n, order=1, act=self._act_code, p0=0.0, p1=0.0, table=_table(self._coeffs(), n))

    def _forward3d(self, x):
        """[B, C, D, H, W] layers (the ...KANConv3DLayer shims of the reference): each depth tap is one launch set of the 2-D kernels (conv3d_stage)."""
        n = self._n_planes()                                     # (Taylor: degree planes, not degree + 1 -- same source as conv_spec())
        kw = dict(kind=L.BASIS_POLY, n_basis=n, order=1, act=self._act_code, p0=0.0, p1=0.0, table=_table(self._coeffs(), n))
        xa, xb = self._base_input(x)
        z = conv3d_stage(kw, self.kernel_size, self.stride, self.padding, self.dilation, self.groups, xa, xb,
                         [m.weight for m in self.base_conv], [m.weight for m in self.poly_conv])
        y = _norm3d(self.layer_norm, self.prelus, z, self.output_dim_group)
        return self.dropout(y) if self.dropout is not None else y

    def forward(self, x, pool=False):
        """`pool` = True or (kernel, stride): max_pool2d(layer(x), ...) with the pooling inside the InstanceNorm+PReLU kernels (see KANConvNDLayer)."""
        if self.ndim == 3:
            if pool:
                raise NotImplementedError("pool=True is a 2-D fusion")
            return self._forward3d(x)
        spec = self.conv_spec()
        x = self._lift(x)
        wb, ws = self._w(self.base_conv), self._w(self.poly_conv)
        prelus = [m.weight for m in self.prelus]
        xa, xb = self._base_input(x)                              # (act(x), x) when the host applies the activation
        if xb is None and _fusable_instnorm(self.layer_norm) and all(p.numel() == 1 for p in prelus):
            gam, bet = self._norm_affine(self.layer_norm)
            if pool and self.dropout is None and self.ndim == 2:
                ho, wo = spec.out_hw(x.shape[2], x.shape[3])
                if pool is not True or (ho % 2 == 0 and wo % 2 == 0):
                    return ops.kan_conv_in_prelu(spec, x, wb, ws, gam, bet, prelus, eps=self.layer_norm[0].eps, pool=pool)
            y = self._lower(ops.kan_conv_in_prelu(spec, x, wb, ws, gam, bet, prelus, eps=self.layer_norm[0].eps))
        else:
            y = self._norm_prelu(ops.kan_conv(spec, xa, xb, wb, ws))
        if self.dropout is not None:
            y = self.dropout(y)
        return _unfused_pool(y, pool) if pool else y


# ------------------------------------------------------------------------------------------- Bessel
class BesselKANConvNDLayer(_RecurrenceKANConvNDLayer):
    """y_0 = 1, y_1 = t + 1, y_n = (2n-1) t y_{n-1} + y_{n-2}  (bessel_kan_layers.py compute_bessel_basis)."""

    def __init__(self, conv_class, norm_class, input_dim, output_dim, kernel_size, degree, groups=1, padding=0, stride=1, dilation=1,
                 ndim: int = 2, base_activation=nn.GELU, dropout: float = 0.0, **norm_kwargs):
        super().__init__()
        self._setup(conv_class, norm_class, input_dim, output_dim, kernel_size, degree, groups, padding, stride, dilation, ndim,
                    base_activation, dropout, norm_kwargs)

    def _coeffs(self):
        return 1.0, 1.0, 1.0, [(2.0 * k - 1.0, 0.0, 1.0) for k in range(2, self.degree + 1)]


class BesselKANConv2DLayer(BesselKANConvNDLayer):
    def __init__(self, input_dim, output_dim, kernel_size, degree, groups=1, padding=0, stride=1, dilation=1,
                 base_activation=nn.GELU, dropout=0.0, norm_layer=nn.InstanceNorm2d, **norm_kwargs):
        super().__init__(conv_class=nn.Conv2d, norm_class=norm_layer, input_dim=input_dim, output_dim=output_dim,
                         kernel_size=kernel_size, degree=degree, groups=groups, padding=padding, stride=stride, dilation=dilation,
                         ndim=2, base_activation=base_activation, dropout=dropout, **norm_kwargs)


class BesselKANConv1DLayer(BesselKANConvNDLayer):
    """bessel_kan_layers.py:192-200: the same layer on [B, C, L] (nn.Conv1d weights [O, C, k], InstanceNorm1d)."""

    def __init__(self, input_dim, output_dim, kernel_size, degree, groups=1, padding=0, stride=1, dilation=1,
                 base_activation=nn.GELU, dropout=0.0, norm_layer=nn.InstanceNorm1d, **norm_kwargs):
        super().__init__(conv_class=nn.Conv1d, norm_class=norm_layer, input_dim=input_dim, output_dim=output_dim,
                         kernel_size=kernel_size, degree=degree, groups=groups, padding=padding, stride=stride, dilation=dilation,
                         ndim=1, base_activation=base_activation, dropout=dropout, **norm_kwargs)


class BesselKANConv3DLayer(BesselKANConvNDLayer):
    """The 3-D shim (bessel_kan_layers.py, ...KANConv3DLayer): [B, C, D, H, W], nn.Conv3d weights, InstanceNorm3d; depth taps around the 2-D kernels."""

    def __init__(self, input_dim, output_dim, kernel_size, degree, groups=1, padding=0, stride=1, dilation=1,
                 base_activation=nn.GELU, dropout=0.0, norm_layer=nn.InstanceNorm3d, **norm_kwargs):
        super().__init__(conv_class=nn.Conv3d, norm_class=norm_layer, input_dim=input_dim, output_dim=output_dim,
                         kernel_size=kernel_size, degree=degree, groups=groups, padding=padding, stride=stride, dilation=dilation,
                         ndim=3, base_activation=base_activation, dropout=dropout, **norm_kwargs)


# ------------------------------------------------------------------------------------------- Fibonacci
class FibonacciKANConvNDLayer(_RecurrenceKANConvNDLayer):
    """F_0 = 0, F_1 = 1, F_n = t F_{n-1} + F_{n-2}  (fibonacci_kan_layers.py compute_fibonacci_basis)."""
    _min_degree = 1
    _min_degree_msg = 'degree must be at least 1'

    def __init__(self, conv_class, norm_class, input_dim, output_dim, kernel_size, degree, groups=1, padding=0, stride=1, dilation=1,
                 ndim: int = 2, base_activation=nn.GELU, dropout: float = 0.0, **norm_kwargs):
        super().__init__()
        self._setup(conv_class, norm_class, input_dim, output_dim, kernel_size, degree, groups, padding, stride, dilation, ndim,
                    base_activation, dropout, norm_kwargs)

    def _coeffs(self):
        return 0.0, 0.0, 1.0, [(1.0, 0.0, 1.0) for _ in range(2, self.degree + 1)]


class FibonacciKANConv2DLayer(FibonacciKANConvNDLayer):
    def __init__(self, input_dim, output_dim, kernel_size, degree, groups=1, padding=0, stride=1, dilation=1,
                 base_activation=nn.GELU, dropout=0.0, norm_layer=nn.InstanceNorm2d, **norm_kwargs):
        super().__init__(conv_class=nn.Conv2d, norm_class=norm_layer, input_dim=input_dim, output_dim=output_dim,
                         kernel_size=kernel_size, degree=degree, groups=groups, padding=padding, stride=stride, dilation=dilation,
                         ndim=2, base_activation=base_activation, dropout=dropout, **norm_kwargs)


class FibonacciKANConv1DLayer(FibonacciKANConvNDLayer):
    """fibonacci_kan_layers.py:245-259: the same layer on [B, C, L] (nn.Conv1d weights [O, C, k], InstanceNorm1d)."""

    def __init__(self, input_dim, output_dim, kernel_size, degree, groups=1, padding=0, stride=1, dilation=1,
                 base_activation=nn.GELU, dropout=0.0, norm_layer=nn.InstanceNorm1d, **norm_kwargs):
        super().__init__(conv_class=nn.Conv1d, norm_class=norm_layer, input_dim=input_dim, output_dim=output_dim,
                         kernel_size=kernel_size, degree=degree, groups=groups, padding=padding, stride=stride, dilation=dilation,
                         ndim=1, base_activation=base_activation, dropout=dropout, **norm_kwargs)


class FibonacciKANConv3DLayer(FibonacciKANConvNDLayer):
    """The 3-D shim (fibonacci_kan_layers.py, ...KANConv3DLayer): [B, C, D, H, W], nn.Conv3d weights, InstanceNorm3d; depth taps around the 2-D kernels."""

    def __init__(self, input_dim, output_dim, kernel_size, degree, groups=1, padding=0, stride=1, dilation=1,
                 base_activation=nn.GELU, dropout=0.0, norm_layer=nn.InstanceNorm3d, **norm_kwargs):
        super().__init__(conv_class=nn.Conv3d, norm_class=norm_layer, input_dim=input_dim, output_dim=output_dim,
                         kernel_size=kernel_size, degree=degree, groups=groups, padding=padding, stride=stride, dilation=dilation,
                         ndim=3, base_activation=base_activation, dropout=dropout, **norm_kwargs)


# ------------------------------------------------------------------------------------------- Gegenbauer
class GegenbauerKANConvNDLayer(_RecurrenceKANConvNDLayer):
    """C_0 = 1, C_1 = 2 a t, C_{n+1} = (2 (n + a) t C_n - (n + 2a - 1) C_{n-1}) / (n + 1)  (gegenbauer_kan_layers.py)."""

    def __init__(self, conv_class, norm_class, input_dim, output_dim, kernel_size, degree, alpha_param, groups=1, padding=0, stride=1,
                 dilation=1, ndim: int = 2, base_activation=nn.GELU, dropout: float = 0.0, **norm_kwargs):
        super().__init__()
        if alpha_param <= -0.5:
            raise ValueError('alpha_param must be greater than -0.5')
        self.alpha_param = alpha_param
        self._setup(conv_class, norm_class, input_dim, output_dim, kernel_size, degree, groups, padding, stride, dilation, ndim,
                    base_activation, dropout, norm_kwargs)

    def _coeffs(self):
        a = float(self.alpha_param)
        return 1.0, 2.0 * a, 0.0, [(2.0 * (k - 1 + a) / k, 0.0, -(k + 2.0 * a - 2.0) / k) for k in range(2, self.degree + 1)]


class GegenbauerKANConv2DLayer(GegenbauerKANConvNDLayer):
    def __init__(self, input_dim, output_dim, kernel_size, degree, alpha_param, groups=1, padding=0, stride=1, dilation=1,
                 base_activation=nn.GELU, dropout=0.0, norm_layer=nn.InstanceNorm2d, **norm_kwargs):
        super().__init__(conv_class=nn.Conv2d, norm_class=norm_layer, input_dim=input_dim, output_dim=output_dim,
                         kernel_size=kernel_size, degree=degree, alpha_param=alpha_param, groups=groups, padding=padding, stride=stride,
                         dilation=dilation, ndim=2, base_activation=base_activation, dropout=dropout, **norm_kwargs)


class GegenbauerKANConv1DLayer(GegenbauerKANConvNDLayer):
    """gegenbauer_kan_layers.py:227-241: the same layer on [B, C, L] (nn.Conv1d weights [O, C, k], InstanceNorm1d)."""

    def __init__(self, input_dim, output_dim, kernel_size, degree, alpha_param, groups=1, padding=0, stride=1, dilation=1,
                 base_activation=nn.GELU, dropout=0.0, norm_layer=nn.InstanceNorm1d, **norm_kwargs):
        super().__init__(conv_class=nn.Conv1d, norm_class=norm_layer, input_dim=input_dim, output_dim=output_dim,
                         kernel_size=kernel_size, degree=degree, alpha_param=alpha_param, groups=groups, padding=padding, stride=stride,
                         dilation=dilation, ndim=1, base_activation=base_activation, dropout=dropout, **norm_kwargs)


class GegenbauerKANConv3DLayer(GegenbauerKANConvNDLayer):
    """The 3-D shim (gegenbauer_kan_layers.py, ...KANConv3DLayer): [B, C, D, H, W], nn.Conv3d weights, InstanceNorm3d; depth taps around the 2-D kernels."""

    def __init__(self, input_dim, output_dim, kernel_size, degree, alpha_param, groups=1, padding=0, stride=1, dilation=1,
                 base_activation=nn.GELU, dropout=0.0, norm_layer=nn.InstanceNorm3d, **norm_kwargs):
        super().__init__(conv_class=nn.Conv3d, norm_class=norm_layer, input_dim=input_dim, output_dim=output_dim,
                         kernel_size=kernel_size, degree=degree, alpha_param=alpha_param, groups=groups, padding=padding, stride=stride,
                         dilation=dilation, ndim=3, base_activation=base_activation, dropout=dropout, **norm_kwargs)


# ------------------------------------------------------------------------------------------- Hermite
class HermiteKANConvNDLayer(_RecurrenceKANConvNDLayer):
    """H_0 = 1, H_1 = 2t, H_n = 2t H_{n-1} - 2(n-1) H_{n-2}  (hermite_kan_layers.py:117-146; base_activation is called as given, :65)."""

    def __init__(self, conv_class, norm_class, input_dim, output_dim, kernel_size, degree, groups=1, padding=0, stride=1, dilation=1,
                 ndim: int = 2, base_activation=nn.GELU, dropout: float = 0.0, **norm_kwargs):
        super().__init__()
        self._setup(conv_class, norm_class, input_dim, output_dim, kernel_size, degree, groups, padding, stride, dilation, ndim,
                    base_activation, dropout, norm_kwargs, none_is_identity=False)

    def _coeffs(self):
        return 1.0, 2.0, 0.0, [(2.0, 0.0, -2.0 * (k - 1)) for k in range(2, self.degree + 1)]


class HermiteKANConv2DLayer(HermiteKANConvNDLayer):
    def __init__(self, input_dim, output_dim, kernel_size, degree, groups=1, padding=0, stride=1, dilation=1,
                 base_activation=nn.GELU, dropout=0.0, norm_layer=nn.InstanceNorm2d, **norm_kwargs):
        super().__init__(conv_class=nn.Conv2d, norm_class=norm_layer, input_dim=input_dim, output_dim=output_dim,
                         kernel_size=kernel_size, degree=degree, groups=groups, padding=padding, stride=stride, dilation=dilation,
                         ndim=2, base_activation=base_activation, dropout=dropout, **norm_kwargs)


class HermiteKANConv1DLayer(HermiteKANConvNDLayer):
    """hermite_kan_layers.py:185-193: the same layer on [B, C, L] (nn.Conv1d weights [O, C, k], InstanceNorm1d)."""

    def __init__(self, input_dim, output_dim, kernel_size, degree, groups=1, padding=0, stride=1, dilation=1,
                 base_activation=nn.GELU, dropout=0.0, norm_layer=nn.InstanceNorm1d, **norm_kwargs):
        super().__init__(conv_class=nn.Conv1d, norm_class=norm_layer, input_dim=input_dim, output_dim=output_dim,
                         kernel_size=kernel_size, degree=degree, groups=groups, padding=padding, stride=stride, dilation=dilation,
                         ndim=1, base_activation=base_activation, dropout=dropout, **norm_kwargs)


class HermiteKANConv3DLayer(HermiteKANConvNDLayer):
    """The 3-D shim (hermite_kan_layers.py, ...KANConv3DLayer): [B, C, D, H, W], nn.Conv3d weights, InstanceNorm3d; depth taps around the 2-D kernels."""

    def __init__(self, input_dim, output_dim, kernel_size, degree, groups=1, padding=0, stride=1, dilation=1,
                 base_activation=nn.GELU, dropout=0.0, norm_layer=nn.InstanceNorm3d, **norm_kwargs):
        super().__init__(conv_class=nn.Conv3d, norm_class=norm_layer, input_dim=input_dim, output_dim=output_dim,
                         kernel_size=kernel_size, degree=degree, groups=groups, padding=padding, stride=stride, dilation=dilation,
                         ndim=3, base_activation=base_activation, dropout=dropout, **norm_kwargs)


# ------------------------------------------------------------------------------------------- Laguerre
class LaguerreKANConvNDLayer(_RecurrenceKANConvNDLayer):
    """L_0 = 1, L_1 = 1 + a - t, k L_k = (2k - 1 + a - t) L_{k-1} - (k - 1 + a) L_{k-2}  (laguerre_kan_layers.py)."""

    def __init__(self, conv_class, norm_class, input_dim, output_dim, kernel_size, degree, alpha, groups=1, padding=0, stride=1,
                 dilation=1, ndim: int = 2, base_activation=nn.GELU, dropout: float = 0.0, **norm_kwargs):
        super().__init__()
        if alpha <= -1.0:
            raise ValueError('alpha must be greater than -1 for Laguerre polynomials')
        self.alpha = alpha
        self._setup(conv_class, norm_class, input_dim, output_dim, kernel_size, degree, groups, padding, stride, dilation, ndim,
                    base_activation, dropout, norm_kwargs)

    def _coeffs(self):
        a = float(self.alpha)
        return 1.0, -1.0, 1.0 + a, [(-1.0 / k, (2.0 * k - 1.0 + a) / k, -(k - 1.0 + a) / k) for k in range(2, self.degree + 1)]


class LaguerreKANConv2DLayer(LaguerreKANConvNDLayer):
    def __init__(self, input_dim, output_dim, kernel_size, degree, alpha, groups=1, padding=0, stride=1, dilation=1,
                 base_activation=nn.GELU, dropout=0.0, norm_layer=nn.InstanceNorm2d, **norm_kwargs):
        super().__init__(conv_class=nn.Conv2d, norm_class=norm_layer, input_dim=input_dim, output_dim=output_dim,
                         kernel_size=kernel_size, degree=degree, alpha=alpha, groups=groups, padding=padding, stride=stride,
                         dilation=dilation, ndim=2, base_activation=base_activation, dropout=dropout, **norm_kwargs)


class LaguerreKANConv1DLayer(LaguerreKANConvNDLayer):
    """laguerre_kan_layers.py:204-212: the same layer on [B, C, L] (nn.Conv1d weights [O, C, k], InstanceNorm1d)."""

    def __init__(self, input_dim, output_dim, kernel_size, degree, alpha, groups=1, padding=0, stride=1, dilation=1,
                 base_activation=nn.GELU, dropout=0.0, norm_layer=nn.InstanceNorm1d, **norm_kwargs):
        super().__init__(conv_class=nn.Conv1d, norm_class=norm_layer, input_dim=input_dim, output_dim=output_dim,
                         kernel_size=kernel_size, degree=degree, alpha=alpha, groups=groups, padding=padding, stride=stride,
                         dilation=dilation, ndim=1, base_activation=base_activation, dropout=dropout, **norm_kwargs)


class LaguerreKANConv3DLayer(LaguerreKANConvNDLayer):
    """The 3-D shim (laguerre_kan_layers.py, ...KANConv3DLayer): [B, C, D, H, W], nn.Conv3d weights, InstanceNorm3d; depth taps around the 2-D kernels."""

    def __init__(self, input_dim, output_dim, kernel_size, degree, alpha, groups=1, padding=0, stride=1, dilation=1,
                 base_activation=nn.GELU, dropout=0.0, norm_layer=nn.InstanceNorm3d, **norm_kwargs):
        super().__init__(conv_class=nn.Conv3d, norm_class=norm_layer, input_dim=input_dim, output_dim=output_dim,
                         kernel_size=kernel_size, degree=degree, alpha=alpha, groups=groups, padding=padding, stride=stride,
                         dilation=dilation, ndim=3, base_activation=base_activation, dropout=dropout, **norm_kwargs)


# ------------------------------------------------------------------------------------------- Lucas
class LucasKANConvNDLayer(_RecurrenceKANConvNDLayer):
    """L_0 = 2, L_1 = t, L_n = t L_{n-1} + L_{n-2}  (lucas_kan_layers.py:140-174)."""

    def __init__(self, conv_class, norm_class, input_dim, output_dim, kernel_size, degree, groups=1, padding=0, stride=1, dilation=1,
                 ndim: int = 2, base_activation=nn.GELU, dropout: float = 0.0, **norm_kwargs):
        super().__init__()
        self._setup(conv_class, norm_class, input_dim, output_dim, kernel_size, degree, groups, padding, stride, dilation, ndim,
                    base_activation, dropout, norm_kwargs)

    def _coeffs(self):
        return 2.0, 1.0, 0.0, [(1.0, 0.0, 1.0) for _ in range(2, self.degree + 1)]


class LucasKANConv2DLayer(LucasKANConvNDLayer):
    def __init__(self, input_dim, output_dim, kernel_size, degree, groups=1, padding=0, stride=1, dilation=1,
                 base_activation=nn.GELU, dropout=0.0, norm_layer=nn.InstanceNorm2d, **norm_kwargs):
        super().__init__(conv_class=nn.Conv2d, norm_class=norm_layer, input_dim=input_dim, output_dim=output_dim,
                         kernel_size=kernel_size, degree=degree, groups=groups, padding=padding, stride=stride, dilation=dilation,
                         ndim=2, base_activation=base_activation, dropout=dropout, **norm_kwargs)


class LucasKANConv1DLayer(LucasKANConvNDLayer):
    """lucas_kan_layers.py:220-228: the same layer on [B, C, L] (nn.Conv1d weights [O, C, k], InstanceNorm1d)."""

    def __init__(self, input_dim, output_dim, kernel_size, degree, groups=1, padding=0, stride=1, dilation=1,
                 base_activation=nn.GELU, dropout=0.0, norm_layer=nn.InstanceNorm1d, **norm_kwargs):
        super().__init__(conv_class=nn.Conv1d, norm_class=norm_layer, input_dim=input_dim, output_dim=output_dim,
                         kernel_size=kernel_size, degree=degree, groups=groups, padding=padding, stride=stride, dilation=dilation,
                         ndim=1, base_activation=base_activation, dropout=dropout, **norm_kwargs)


class LucasKANConv3DLayer(LucasKANConvNDLayer):
    """The 3-D shim (lucas_kan_layers.py, ...KANConv3DLayer): [B, C, D, H, W], nn.Conv3d weights, InstanceNorm3d; depth taps around the 2-D kernels."""

    def __init__(self, input_dim, output_dim, kernel_size, degree, groups=1, padding=0, stride=1, dilation=1,
                 base_activation=nn.GELU, dropout=0.0, norm_layer=nn.InstanceNorm3d, **norm_kwargs):
        super().__init__(conv_class=nn.Conv3d, norm_class=norm_layer, input_dim=input_dim, output_dim=output_dim,
                         kernel_size=kernel_size, degree=degree, groups=groups, padding=padding, stride=stride, dilation=dilation,
                         ndim=3, base_activation=base_activation, dropout=dropout, **norm_kwargs)


# ------------------------------------------------------------------------------------------- Taylor
class TaylorKANConvNDLayer(_RecurrenceKANConvNDLayer):
    """`degree` monomials t^0 .. t^(degree-1)  (taylor_kan_layers.py compute_taylor_basis)."""
    _min_degree = 1
    _min_degree_msg = 'degree must be at least 1'

    def __init__(self, conv_class, norm_class, input_dim, output_dim, kernel_size, degree, groups=1, padding=0, stride=1, dilation=1,
                 ndim: int = 2, base_activation=nn.GELU, dropout: float = 0.0, **norm_kwargs):
        super().__init__()
        self._setup(conv_class, norm_class, input_dim, output_dim, kernel_size, degree, groups, padding, stride, dilation, ndim,
                    base_activation, dropout, norm_kwargs)

    def _n_planes(self):
        return self.degree

    def _coeffs(self):
        return 1.0, 1.0, 0.0, [(1.0, 0.0, 0.0) for _ in range(2, self.degree)]


class TaylorKANConv2DLayer(TaylorKANConvNDLayer):
    def __init__(self, input_dim, output_dim, kernel_size, degree, groups=1, padding=0, stride=1, dilation=1,
                 base_activation=nn.GELU, dropout=0.0, norm_layer=nn.InstanceNorm2d, **norm_kwargs):
        super().__init__(conv_class=nn.Conv2d, norm_class=norm_layer, input_dim=input_dim, output_dim=output_dim,
                         kernel_size=kernel_size, degree=degree, groups=groups, padding=padding, stride=stride, dilation=dilation,
                         ndim=2, base_activation=base_activation, dropout=dropout, **norm_kwargs)


class TaylorKANConv1DLayer(TaylorKANConvNDLayer):
    """taylor_kan_layers.py:197-205: the same layer on [B, C, L] (nn.Conv1d weights [O, C, k], InstanceNorm1d)."""

    def __init__(self, input_dim, output_dim, kernel_size, degree, groups=1, padding=0, stride=1, dilation=1,
                 base_activation=nn.GELU, dropout=0.0, norm_layer=nn.InstanceNorm1d, **norm_kwargs):
        super().__init__(conv_class=nn.Conv1d, norm_class=norm_layer, input_dim=input_dim, output_dim=output_dim,
                         kernel_size=kernel_size, degree=degree, groups=groups, padding=padding, stride=stride, dilation=dilation,
                         ndim=1, base_activation=base_activation, dropout=dropout, **norm_kwargs)


class TaylorKANConv3DLayer(TaylorKANConvNDLayer):
    """The 3-D shim (taylor_kan_layers.py, ...KANConv3DLayer): [B, C, D, H, W], nn.Conv3d weights, InstanceNorm3d; depth taps around the 2-D kernels."""

    def __init__(self, input_dim, output_dim, kernel_size, degree, groups=1, padding=0, stride=1, dilation=1,
                 base_activation=nn.GELU, dropout=0.0, norm_layer=nn.InstanceNorm3d, **norm_kwargs):
        super().__init__(conv_class=nn.Conv3d, norm_class=norm_layer, input_dim=input_dim, output_dim=output_dim,
                         kernel_size=kernel_size, degree=degree, groups=groups, padding=padding, stride=stride, dilation=dilation,
                         ndim=3, base_activation=base_activation, dropout=dropout, **norm_kwargs)


# ------------------------------------------------------------------------------------------- Fourier
class FourierKANConvNDLayer(_HipLayer):
    """fourier_kan_layers.py:63-212: y = Dropout(PReLU(norm(conv(act(x), W_base) + conv([cos(kx)]_k ++ [sin(kx)]_k, W_fourier)))),
    k = 1..grid_size, channel index c*2G + (k-1) for the cosines and c*2G + G + (k-1) for the sines (:184-186)."""

    def __init__(self, conv_class, norm_class, input_dim, output_dim, kernel_size, grid_size, groups=1, padding=0, stride=1, dilation=1,
                 ndim: int = 2, base_activation=nn.GELU, dropout: float = 0.0, smooth_initialization: bool = False, **norm_kwargs):
        super().__init__()
        _need_conv1d_or_2d(conv_class, ndim, allow_3d=True)
        _check_groups(groups, input_dim, output_dim)
        if grid_size < 1:
            raise ValueError('grid_size must be at least 1')
        if 2 * grid_size + 1 > L.KAN_MAX_PLANES:
            raise NotImplementedError(f"the HIP conv stage holds at most {L.KAN_MAX_PLANES} planes per channel (grid_size <= 7)")
        self.input_dim, self.output_dim, self.kernel_size, self.grid_size = input_dim, output_dim, kernel_size, grid_size
        self.groups, self.padding, self.stride, self.dilation, self.ndim = groups, padding, stride, dilation, ndim
        self.base_activation = base_activation() if base_activation is not None else nn.Identity()
        self.norm_kwargs = norm_kwargs
        self.input_dim_group, self.output_dim_group = input_dim // groups, output_dim // groups
        self.fourier_input_dim_group = self.input_dim_group * (2 * grid_size)
        cg, og = self.input_dim_group, self.output_dim_group
        self.base_conv = nn.ModuleList([conv_class(cg, og, kernel_size, stride, padding, dilation, groups=1, bias=False)
                                        for _ in range(groups)])
        self.fourier_conv = nn.ModuleList([conv_class(self.fourier_input_dim_group, og, kernel_size, stride, padding, dilation, groups=1,
                                                      bias=False) for _ in range(groups)])
        self.layer_norm = nn.ModuleList([norm_class(og, **_filter_norm_kwargs(norm_class, norm_kwargs)) for _ in range(groups)])
        self.prelus = nn.ModuleList([nn.PReLU() for _ in range(groups)])
        self.dropout = _dropout2d(dropout, ndim)
        for conv in self.base_conv:
            nn.init.kaiming_uniform_(conv.weight, nonlinearity='linear')
        for conv in self.fourier_conv:
            nn.init.kaiming_uniform_(conv.weight, nonlinearity='linear')
        self._act_code = _act_code(self.base_activation, host_ok=True)

    def conv_spec(self) -> ops.ConvSpec:
        return self._spec(kind=L.BASIS_FOURIER, n_basis=2 * self.grid_size, order=0, act=self._act_code, p0=0.0, p1=0.0, table=())

    def _forward3d(self, x):
        kw = dict(kind=L.BASIS_FOURIER, n_basis=2 * self.grid_size, order=0, act=self._act_code, p0=0.0, p1=0.0, table=())
        xa, xb = self._base_input(x)
        z = conv3d_stage(kw, self.kernel_size, self.stride, self.padding, self.dilation, self.groups, xa, xb,
                         [m.weight for m in self.base_conv], [m.weight for m in self.fourier_conv])
        y = _norm3d(self.layer_norm, self.prelus, z, self.output_dim_group)
        return self.dropout(y) if self.dropout is not None else y

    def forward(self, x):
        if self.ndim == 3:
            return self._forward3d(x)
        spec = self.conv_spec()
        x = self._lift(x)
        wb, ws = self._w(self.base_conv), self._w(self.fourier_conv)
        prelus = [m.weight for m in self.prelus]
        xa, xb = self._base_input(x)                              # (act(x), x) when the host applies the activation
        if xb is None and _fusable_instnorm(self.layer_norm) and all(p.numel() == 1 for p in prelus):
            gam, bet = self._norm_affine(self.layer_norm)
            y = self._lower(ops.kan_conv_in_prelu(spec, x, wb, ws, gam, bet, prelus, eps=self.layer_norm[0].eps))
        else:
            y = self._norm_prelu(ops.kan_conv(spec, xa, xb, wb, ws))
        if self.dropout is not None:
            y = self.dropout(y)
        return y


class FourierKANConv2DLayer(FourierKANConvNDLayer):
    def __init__(self, input_dim, output_dim, kernel_size, grid_size, groups=1, padding=0, stride=1, dilation=1,
                 base_activation=nn.GELU, dropout=0.0, norm_layer=nn.InstanceNorm2d, **norm_kwargs):
        super().__init__(conv_class=nn.Conv2d, norm_class=norm_layer, input_dim=input_dim, output_dim=output_dim,
                         kernel_size=kernel_size, grid_size=grid_size, groups=groups, padding=padding, stride=stride, dilation=dilation,
                         ndim=2, base_activation=base_activation, dropout=dropout, **norm_kwargs)


class FourierKANConv1DLayer(FourierKANConvNDLayer):
    """fourier_kan_layers.py:233-241: the same layer on [B, C, L] (nn.Conv1d weights [O, C, k], InstanceNorm1d)."""

    def __init__(self, input_dim, output_dim, kernel_size, grid_size, groups=1, padding=0, stride=1, dilation=1,
                 base_activation=nn.GELU, dropout=0.0, norm_layer=nn.InstanceNorm1d, **norm_kwargs):
        super().__init__(conv_class=nn.Conv1d, norm_class=norm_layer, input_dim=input_dim, output_dim=output_dim,
                         kernel_size=kernel_size, grid_size=grid_size, groups=groups, padding=padding, stride=stride, dilation=dilation,
                         ndim=1, base_activation=base_activation, dropout=dropout, **norm_kwargs)


class FourierKANConv3DLayer(FourierKANConvNDLayer):
    """The 3-D shim (fourier_kan_layers.py, FourierKANConv3DLayer): [B, C, D, H, W], nn.Conv3d weights, InstanceNorm3d; depth taps around the 2-D kernels."""

    def __init__(self, input_dim, output_dim, kernel_size, grid_size, groups=1, padding=0, stride=1, dilation=1,
                 base_activation=nn.GELU, dropout=0.0, norm_layer=nn.InstanceNorm3d, **norm_kwargs):
        super().__init__(conv_class=nn.Conv3d, norm_class=norm_layer, input_dim=input_dim, output_dim=output_dim,
                         kernel_size=kernel_size, grid_size=grid_size, groups=groups, padding=padding, stride=stride, dilation=dilation,
                         ndim=3, base_activation=base_activation, dropout=dropout, **norm_kwargs)


# ------------------------------------------------------------------------------------------- Jacobi
class JacobiKANConvNDLayer(_HipLayer):
    """jacobi_kan_layers.py:55-175: y = act(norm(conv(x, W_base) + conv(P(tanh x), poly_weights[g]))), the polynomial planes
    concatenated PLANE-major (channel index k*C + c, :136) and the activation applied after the norm (:165)."""

    def __init__(self, conv_class, norm_class, conv_w_fun, input_dim, output_dim, degree, kernel_size, base_activation=nn.GELU,
                 a: float = 1.0, b: float = 1.0, groups=1, padding=0, stride=1, dilation=1, dropout: float = 0.0, ndim: int = 2,
                 **norm_kwargs):
        super().__init__()
        _need_conv2d(conv_class, ndim)
        self.input_dim, self.output_dim, self.degree, self.kernel_size = input_dim, output_dim, degree, kernel_size
        self.padding, self.stride, self.dilation, self.groups = padding, stride, dilation, groups
        self.base_activation = base_activation() if base_activation is not None else nn.Identity()
        self.conv_w_fun, self.ndim, self.norm_kwargs, self.a, self.b = conv_w_fun, ndim, norm_kwargs, a, b
        self.dropout = None
        if dropout > 0:
            raise NotImplementedError("JacobiKAN applies dropout to the expanded basis planes (jacobi_kan_layers.py:148-149); "
                                      "the fused conv stage never materialises them -- use dropout=0")
        _check_groups(groups, input_dim, output_dim)
        if degree < 1 or degree > 10:
            raise NotImplementedError("JacobiKAN on the HIP path needs 1 <= degree <= 10")
        if not isinstance(kernel_size, int):
            raise TypeError("JacobiKAN takes an int kernel_size (jacobi_kan_layers.py:108-109,116)")
        cg, og = input_dim // groups, output_dim // groups
        self.base_conv = nn.ModuleList([conv_class(cg, og, kernel_size, stride, padding, dilation, groups=1, bias=False)
                                        for _ in range(groups)])
        self.layer_norm = nn.ModuleList([norm_class(og, **_filter_norm_kwargs(norm_class, norm_kwargs)) for _ in range(groups)])
        self.poly_weights = nn.Parameter(torch.randn(groups, og, cg * (degree + 1), *([kernel_size] * ndim)))
        for conv in self.base_conv:
            nn.init.kaiming_uniform_(conv.weight, nonlinearity='linear')
        nn.init.normal_(self.poly_weights, mean=0.0, std=1 / (input_dim * (degree + 1) * kernel_size ** ndim))

    def _coeffs(self) -> Coeffs:
        a, b = float(self.a), float(self.b)
        rec = []
        for i in range(2, self.degree + 1):                      # jacobi_kan_layers.py:127-133
            th = (2 * i + a + b) * (2 * i + a + b - 1) / (2 * i * (i + a + b))
            th1 = (2 * i + a + b - 1) * (a * a - b * b) / (2 * i * (i + a + b) * (2 * i + a + b - 2))
            th2 = (i + a - 1) * (i + b - 1) * (2 * i + a + b) / (i * (i + a + b) * (2 * i + a + b - 2))
            rec.append((th, th1, -th2))
        return 1.0, (a + b + 2.0) / 2.0, (a - b) / 2.0, rec

    def conv_spec(self) -> ops.ConvSpec:
        n = self.degree + 1
        return self._spec(kind=L.BASIS_POLY, n_basis=n, order=1, act=L.ACT_IDENTITY, p0=0.0, p1=0.0, table=_table(self._coeffs(), n))

    def forward(self, x):
        G, n = self.groups, self.degree + 1
        og, cg, k = self.output_dim // G, self.input_dim // G, self.kernel_size
        # plane-major (k*C + c) -> the kernels' channel-major (c*n + k) order; autograd carries the gradient back
        ws = [self.poly_weights[g].view(og, n, cg, k, k).transpose(1, 2).reshape(og, cg * n, k, k) for g in range(G)]
        z = ops.kan_conv(self.conv_spec(), x, None, [m.weight for m in self.base_conv], ws)
        if _fusable_instnorm(self.layer_norm):
            gam, bet = self._norm_affine(self.layer_norm)
            y = ops.instance_norm(z, torch.cat(gam) if gam is not None else None, torch.cat(bet) if bet is not None else None,
                                  eps=self.layer_norm[0].eps)
        else:
            parts = []
            for g in range(G):
                zg = z[:, g * og:(g + 1) * og]
                if isinstance(self.layer_norm[g], nn.LayerNorm):
                    zg = self.layer_norm[g](zg.reshape(zg.shape[0], -1)).view(zg.shape)
                else:
                    zg = self.layer_norm[g](zg)
                parts.append(zg)
            y = torch.cat(parts, dim=1)
        return self.base_activation(y)


class JacobiKANConv2DLayer(JacobiKANConvNDLayer):
    def __init__(self, input_dim, output_dim, kernel_size, degree=3, base_activation=nn.GELU, a=1.0, b=1.0, groups=1, padding=0,
                 stride=1, dilation=1, dropout: float = 0.0, norm_layer=nn.InstanceNorm2d, **norm_kwargs):
        super().__init__(conv_class=nn.Conv2d, norm_class=norm_layer, conv_w_fun=torch.nn.functional.conv2d, input_dim=input_dim,
                         output_dim=output_dim, degree=degree, kernel_size=kernel_size, base_activation=base_activation, a=a, b=b,
                         groups=groups, padding=padding, stride=stride, dilation=dilation, ndim=2, dropout=dropout, **norm_kwargs)


# ------------------------------------------------------------------------------------------- Legendre / Bernstein
class _PlaneMajorPolyLayer(_HipLayer):
    """Common body of the 'torchkan-style' conv layers (legendre / bersnstein / jacobi _kan_layers.py): identity base branch,
    one ``poly_weights`` parameter [G, O/G, C/G*(degree+1), k, k], output = base_activation(norm(base + poly))."""

    def _setup_pm(self, conv_class, norm_class, conv_w_fun, input_dim, output_dim, degree, kernel_size, base_activation, groups, padding,
                  stride, dilation, dropout, ndim, norm_kwargs):
        _need_conv2d(conv_class, ndim)
        self.degree, self.kernel_size = degree, kernel_size
        self.padding, self.stride, self.dilation, self.groups = padding, stride, dilation, groups
        self.base_activation = base_activation
        self.conv_w_fun, self.ndim, self.norm_kwargs = conv_w_fun, ndim, norm_kwargs
        self.dropout = _dropout2d(dropout)
        _check_groups(groups, input_dim, output_dim)
        if degree < 1 or degree > 10:
            raise NotImplementedError("this layer on the HIP path needs 1 <= degree <= 10")
        if not isinstance(kernel_size, int):
            raise TypeError("an int kernel_size is required (the reference builds poly_weights from `kernel_size` repeated ndim times)")
        cg, og = input_dim // groups, output_dim // groups
        self.base_conv = nn.ModuleList([conv_class(cg, og, kernel_size, stride, padding, dilation, groups=1, bias=False)
                                        for _ in range(groups)])
        self.layer_norm = nn.ModuleList([norm_class(og, **_filter_norm_kwargs(norm_class, norm_kwargs)) for _ in range(groups)])
        self.poly_weights = nn.Parameter(torch.randn(groups, og, cg * (degree + 1), *([kernel_size] * ndim)))
        for conv in self.base_conv:
            nn.init.kaiming_uniform_(conv.weight, nonlinearity='linear')
        nn.init.kaiming_uniform_(self.poly_weights, nonlinearity='linear')

    def _norm_act(self, z, og):
        G = self.groups
        if _fusable_instnorm(self.layer_norm):
            gam, bet = self._norm_affine(self.layer_norm)
            y = ops.instance_norm(z, torch.cat(gam) if gam is not None else None, torch.cat(bet) if bet is not None else None,
                                  eps=self.layer_norm[0].eps)
        else:
            parts = []
            for g in range(G):
                zg = z[:, g * og:(g + 1) * og]
                if isinstance(self.layer_norm[g], nn.LayerNorm):
                    zg = self.layer_norm[g](zg.reshape(zg.shape[0], -1)).view(zg.shape)
                else:
                    zg = self.layer_norm[g](zg)
                parts.append(zg)
            y = torch.cat(parts, dim=1)
        return self.base_activation(y)


class LegendreKANConvNDLayer(_PlaneMajorPolyLayer):
    """legendre_kan_layers.py:50-158: base_conv(x) + conv(P_0..P_degree(x_n), poly_weights[g]) -> norm -> SiLU, with
    x_n = 2 (x - min) / (max - min) - 1 over the WHOLE group tensor (:130; batch-coupled, so not data-parallel invariant),
    dropout on x_n (:132-133), planes concatenated plane-major k*C + c (:124).  The normalisation is two torch reductions
    (autograd carries the min / max gradients); the HIP conv stage evaluates the recurrence on x_n as a second input."""

    def __init__(self, conv_class, norm_class, conv_w_fun, input_dim, output_dim, degree, kernel_size, groups=1, padding=0, stride=1,
                 dilation=1, dropout: float = 0.0, ndim: int = 2, **norm_kwargs):
        super().__init__()
        self.input_dim, self.output_dim = input_dim, output_dim
        self._setup_pm(conv_class, norm_class, conv_w_fun, input_dim, output_dim, degree, kernel_size, nn.SiLU(), groups, padding, stride,
                       dilation, dropout, ndim, norm_kwargs)

    def conv_spec(self) -> ops.ConvSpec:
        n = self.degree + 1     # P_{k} = ((2k-1) x P_{k-1} - (k-1) P_{k-2}) / k   (legendre_kan_layers.py:119-122)
        c = (1.0, 1.0, 0.0, [((2.0 * k - 1.0) / k, 0.0, -(k - 1.0) / k) for k in range(2, n)])
        return self._spec(kind=L.BASIS_POLY, n_basis=n, order=0, act=L.ACT_IDENTITY, p0=0.0, p1=0.0, table=_table(c, n))

    def forward(self, x):
        G, n = self.groups, self.degree + 1
        og, cg, k = self.output_dim // G, self.input_dim // G, self.kernel_size
        B, _, H, W = x.shape
        xg = x.reshape(B, G, cg, H, W)
        lo, hi = xg.amin(dim=(0, 2, 3, 4), keepdim=True), xg.amax(dim=(0, 2, 3, 4), keepdim=True)
        xn = (2 * (xg - lo) / (hi - lo) - 1).reshape(B, G * cg, H, W)
        if self.dropout is not None:
            xn = self.dropout(xn)
        ws = [self.poly_weights[g].view(og, n, cg, k, k).transpose(1, 2).reshape(og, cg * n, k, k) for g in range(G)]
        z = ops.kan_conv(self.conv_spec(), x, xn.contiguous(), [m.weight for m in self.base_conv], ws)
        return self._norm_act(z, og)


class LegendreKANConv2DLayer(LegendreKANConvNDLayer):
    def __init__(self, input_dim, output_dim, kernel_size, degree=3, groups=1, padding=0, stride=1, dilation=1,
                 dropout: float = 0.0, norm_layer=nn.InstanceNorm2d, **norm_kwargs):
        super().__init__(nn.Conv2d, norm_layer, torch.nn.functional.conv2d, input_dim, output_dim, degree, kernel_size,
                         groups=groups, padding=padding, stride=stride, dilation=dilation, ndim=2, dropout=dropout, **norm_kwargs)


class BersnsteinKANConvNDLayer(_PlaneMajorPolyLayer):
    """bersnstein_kan_layers.py:63-179.  The reference's de Casteljau loop starts from ALL-ONES coefficients (:122), and
    b_i (1 - t) + b_{i+1} t with b = 1 stays 1: every one of its degree+1 'Bernstein' planes is the constant 1 (to an ulp),
    with zero gradient w.r.t. x.  This layer reproduces exactly that: base_conv(x) + conv(ones, poly_weights[g]) -> norm ->
    base_activation, channel index c*(degree+1) + k (:135-136).  Dropout on sigmoid(x) (:148-149) cannot change a constant
    basis and is therefore a no-op here as it is there."""

    def __init__(self, conv_class, norm_class, conv_w_fun, input_dim, output_dim, degree, kernel_size, base_activation=nn.SiLU, groups=1,
                 padding=0, stride=1, dilation=1, dropout: float = 0.0, ndim: int = 2, **norm_kwargs):
        super().__init__()
        self.inputdim, self.outdim = input_dim, output_dim
        self._setup_pm(conv_class, norm_class, conv_w_fun, input_dim, output_dim, degree, kernel_size,
                       base_activation() if base_activation is not None else nn.Identity(), groups, padding, stride, dilation, dropout,
                       ndim, norm_kwargs)

    def conv_spec(self) -> ops.ConvSpec:
        n = self.degree + 1
        c = (1.0, 0.0, 1.0, [(0.0, 1.0, 0.0) for _ in range(2, n)])          # T_k == 1, dT_k/dx == 0
        return self._spec(kind=L.BASIS_POLY, n_basis=n, order=1, act=L.ACT_IDENTITY, p0=0.0, p1=0.0, table=_table(c, n))

    def forward(self, x):
        G = self.groups
        og = self.outdim // G
        z = ops.kan_conv(self.conv_spec(), x, None, [m.weight for m in self.base_conv], [self.poly_weights[g] for g in range(G)])
        return self._norm_act(z, og)


class BersnsteinKANConv2DLayer(BersnsteinKANConvNDLayer):
    def __init__(self, input_dim, output_dim, kernel_size, base_activation=nn.SiLU, degree=3, groups=1, padding=0, stride=1, dilation=1,
                 dropout: float = 0.0, norm_layer=nn.InstanceNorm2d, **norm_kwargs):
        super().__init__(nn.Conv2d, norm_layer, torch.nn.functional.conv2d, input_dim, output_dim, degree, kernel_size,
                         base_activation=base_activation, groups=groups, padding=padding, stride=stride, dilation=dilation, ndim=2,
                         dropout=dropout, **norm_kwargs)
