"""Drop-in conv-KAN layers: same constructor signatures, attribute names, parameter order and
state_dict keys as the reference classes, with forward/backward on the libkanconv HIP kernels.

  reference class                                   this file
  layers/kan_layers.py:116-258  KANConvNDLayer       KANConvNDLayer
  layers/kan_layers.py:274-284  KANConv2DLayer       KANConv2DLayer
  layers/fast_kan_layers.py:34-120 / :137-148        FastKANConvNDLayer / FastKANConv2DLayer
  layers/cheby_kan_layers.py:39-111 / :124-131       ChebyKANConvNDLayer / ChebyKANConv2DLayer
  utils/utils.py:19-33          RadialBasisFunction  RadialBasisFunction

The ``nn.Conv2d`` children are weight holders only (their forward is never called): keeping them
preserves ``state_dict`` keys (``base_conv.0.weight`` ...), lets ``load_state_dict`` move weights
between this package and the reference in both directions, and keeps callers that walk
``.modules()`` for ``nn.Conv2d`` (models/kan_alexnet.py:236-241) working.

The 2-D layers are the hot path.  The 1-D shims (KANConv1DLayer, FastKANConv1DLayer, ChebyKANConv1DLayer:
kan_layers.py:287-297, fast_kan_layers.py:151-162, cheby_kan_layers.py:134-141) run on the same kernels by viewing
[B, C, L] as [B, C, 1, L] with a (1, k) kernel; the 3-D shims (kan_layers.py:261-271 and siblings) run one 2-D launch set per
depth tap (conv3d_stage).
"""
from __future__ import annotations

from inspect import signature
from typing import Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L
from .. import ops


def _pair(v):
    if isinstance(v, (tuple, list)):
        if len(v) != 2:
            raise ValueError(f"expected an int or a pair, got {v!r}")
        return int(v[0]), int(v[1])
    return int(v), int(v)


_ACT_CODES = {nn.Identity: L.ACT_IDENTITY, nn.SiLU: L.ACT_SILU, nn.ReLU: L.ACT_RELU, nn.Tanh: L.ACT_TANH, nn.Sigmoid: L.ACT_SIGMOID}


def _act_code(module: nn.Module, host_ok: bool = False) -> int:
    """Map the instantiated ``base_activation`` module to a kernel activation id.  A module without a device functor is
    applied by the host (``_HipLayer._base_input``) where the layer supports that (`host_ok`): the kernels then see the
    activated tensor as their base-branch input with the identity functor, and the raw input as the basis tensor."""
    if type(module) is nn.GELU:
        return L.ACT_GELU_TANH if getattr(module, "approximate", "none") == "tanh" else L.ACT_GELU
    code = _ACT_CODES.get(type(module))
    if code is None:
        if host_ok:
            return L.ACT_IDENTITY
        raise NotImplementedError(
            f"base_activation {type(module).__name__} has no HIP functor yet (supported: Identity/None, GELU, SiLU, ReLU, Tanh, Sigmoid)")
    return code


def _host_applied(module: nn.Module) -> bool:
    return type(module) is not nn.GELU and type(module) not in _ACT_CODES


def _unfused_pool(y, pool):
    """max_pool2d for the `pool` argument of the layers' forward: True = (2, 2), or a (kernel, stride) pair."""
    k, st = (2, 2) if pool is True else (int(pool[0]), int(pool[1]))
    return F.max_pool2d(y, k, st)


def _dropout2d(p: float, ndim: int = 2):
    if p <= 0:
        return None
    return nn.Dropout1d(p=p) if ndim == 1 else nn.Dropout3d(p=p) if ndim == 3 else nn.Dropout2d(p=p)


def _check_groups(groups, input_dim, output_dim):
    # kan_layers.py:148-153 (same messages in the sibling classes)
    if groups <= 0:
        raise ValueError('groups must be a positive integer')
    if input_dim % groups != 0:
        raise ValueError('input_dim must be divisible by groups')
    if output_dim % groups != 0:
        raise ValueError('output_dim must be divisible by groups')


def _filter_norm_kwargs(norm_class, norm_kwargs):
    valid = signature(norm_class).parameters          # kan_layers.py:178-179
    return {k: v for k, v in norm_kwargs.items() if k in valid}


def _fusable_instnorm(mods) -> bool:
    """True when every per-group norm is a plain InstanceNorm1d/2d (instance statistics in train and eval)."""
    return all(type(m) in (nn.InstanceNorm2d, nn.InstanceNorm1d) and not m.track_running_stats for m in mods)


def _need_conv2d(conv_class, ndim, allow_3d: bool = False):
    ok = ((nn.Conv2d, 2), (nn.Conv1d, 1)) + (((nn.Conv3d, 3),) if allow_3d else ())
    if (conv_class, ndim) not in ok:
        raise NotImplementedError("this layer is built for " + ("1-D, 2-D and 3-D" if allow_3d else "1-D and 2-D") + " on the HIP path")


def _triple(v):
    if isinstance(v, (tuple, list)):
        if len(v) != 3:
            raise ValueError(f"expected an int or a triple, got {v!r}")
        return int(v[0]), int(v[1]), int(v[2])
    return int(v), int(v), int(v)


def conv3d_stage(basis_kw, kernel_size, stride, padding, dilation, groups, x, xn, w_base, w_basis):
    """The fused conv stage for 3-D layers (kan_layers.py:261-271 and the sibling 3-D shims), on the 2-D kernels.

    A 3-D cross-correlation is a sum over the kd depth taps of 2-D ones: output slice `do` takes, for tap `td`, the 2-D
    conv stage of input slice `di = do*sd - pd + td*dd` with the kernel slice W[:, :, td] -- and nothing where `di` falls into
    the depth padding, which is exactly the reference's zero padding of the EXPANDED operand.  So each depth tap is one
    launch set of the 2-D kernels over the batch of all (image, valid depth slice) pairs; torch only gathers the slices and
    sums the kd partial results.  x: [B, C, D, H, W]; weights per group [Og, Cg(*n), kd, kh, kw]."""
    (kd, kh, kw), (sd, sh, sw), (pd, ph, pw), (dd, dh, dw) = _triple(kernel_size), _triple(stride), _triple(padding), _triple(dilation)
    if x.dim() != 5:
        raise ValueError(f"expected [B, C, D, H, W] input for a 3-D layer, got shape {tuple(x.shape)}")
    B, C, D, H, W = x.shape
    Do = (D + 2 * pd - dd * (kd - 1) - 1) // sd + 1
    if Do <= 0:
        raise L.KanConvError(f"empty output depth for input depth {D}")
    spec = ops.ConvSpec(kernel=(kh, kw), stride=(sh, sw), padding=(ph, pw), dilation=(dh, dw), groups=groups, **basis_kw)
    z = None
    for td in range(kd):
        dos = [o for o in range(Do) if 0 <= o * sd - pd + td * dd < D]
        if not dos:
            continue
        d0, n = dos[0] * sd - pd + td * dd, len(dos)              # valid outputs are consecutive; their inputs step by sd

        def gather(t):
            return t[:, :, d0:d0 + (n - 1) * sd + 1:sd].permute(0, 2, 1, 3, 4).reshape(B * n, C, H, W)
        z2 = ops.kan_conv(spec, gather(x), gather(xn) if xn is not None else None,
                          [w[:, :, td] for w in w_base], [w[:, :, td] for w in w_basis])
        z2 = z2.view(B, n, *z2.shape[1:]).permute(0, 2, 1, 3, 4)                      # [B, O, n, Ho, Wo]
        z2 = F.pad(z2, (0, 0, 0, 0, dos[0], Do - dos[0] - n))
        z = z2 if z is None else z + z2
    if z is None:                                                 # every tap of every output lies in the depth padding: all zeros
        ho, wo = spec.out_hw(H, W)
        z = x.new_zeros((B, w_basis[0].shape[0] * groups, Do, ho, wo))
        z = z + 0.0 * (x.sum() + sum(w.sum() for w in list(w_base) + list(w_basis)))      # zero gradients, as autograd gives the reference
    return z


def _norm3d(mods, prelus, z, og):
    """Per-group norm (+ PReLU) of a [B, O, D, H, W] tensor: plain InstanceNorm3d runs on the InstanceNorm kernel with the
    volume as one plane; anything else is the caller's own module."""
    if all(type(m) is nn.InstanceNorm3d and not m.track_running_stats for m in mods):
        B, O, Dz, Hz, Wz = z.shape
        gam = torch.cat([m.weight for m in mods]) if mods[0].affine else None
        bet = torch.cat([m.bias for m in mods]) if mods[0].affine else None
        y = ops.instance_norm(z.reshape(B, O, Dz * Hz, Wz), gam, bet, eps=mods[0].eps).view(B, O, Dz, Hz, Wz)
        parts = [y[:, g * og:(g + 1) * og] for g in range(len(mods))]
    else:
        parts = [mods[g](z[:, g * og:(g + 1) * og]) for g in range(len(mods))]
    if prelus is not None:
        parts = [prelus[g](p) for g, p in enumerate(parts)]
    return parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)


def _one(v):
    if isinstance(v, (tuple, list)):
        if len(v) != 1:
            raise ValueError(f"expected an int or a 1-tuple, got {v!r}")
        return int(v[0])
    return int(v)


class _HipLayer(nn.Module):
    """Shared plumbing.  1-D layers (ndim == 1) are lifted to 2-D: x [B,C,L] -> [B,C,1,L], weights [O,C,k] -> [O,C,1,k]
    (views), kernel (1,k), stride (1,s), padding (0,p), dilation (1,d); InstanceNorm1d over L == InstanceNorm2d over 1 x L."""

    def _spec(self, **kw) -> ops.ConvSpec:
        if getattr(self, "ndim", 2) == 1:
            return ops.ConvSpec(kernel=(1, _one(self.kernel_size)), stride=(1, _one(self.stride)), padding=(0, _one(self.padding)),
                                dilation=(1, _one(self.dilation)), groups=self.groups, **kw)
        return ops.ConvSpec(kernel=_pair(self.kernel_size), stride=_pair(self.stride), padding=_pair(self.padding),
                            dilation=_pair(self.dilation), groups=self.groups, **kw)

    def _lift(self, x):
        if getattr(self, "ndim", 2) == 1:
            if x.dim() != 3:
                raise ValueError(f"expected [B, C, L] input for a 1-D layer, got shape {tuple(x.shape)}")
            return x.unsqueeze(2)
        return x

    def _lower(self, y):
        return y.squeeze(2) if getattr(self, "ndim", 2) == 1 else y

    def _w(self, convs):
        return [m.weight.unsqueeze(2) for m in convs] if getattr(self, "ndim", 2) == 1 else [m.weight for m in convs]

    @staticmethod
    def _norm_affine(mods):
        if mods[0].affine:
            return [m.weight for m in mods], [m.bias for m in mods]
        return None, None

    def _base_input(self, x):
        """(base-branch tensor, basis tensor or None): `(act(x), x)` when the host applies the activation (no device functor
        for this module; the kernels run their two-input form with the identity functor), else `(x, None)`."""
        if _host_applied(self.base_activation):
            return self.base_activation(x), x
        return x, None

    def _norm_prelu(self, z):
        """Un-fused tail for the (lifted) [B, O, H, W] pre-norm tensor: the InstanceNorm kernel (or the caller's own norm
        modules on the layer's own rank), then PReLU.  Returns the tensor in the layer's rank."""
        og, mods = self.output_dim_group, self.layer_norm
        if _fusable_instnorm(mods):
            gam = torch.cat([m.weight for m in mods]) if mods[0].affine else None
            bet = torch.cat([m.bias for m in mods]) if mods[0].affine else None
            n = self._lower(ops.instance_norm(z.contiguous(), gam, bet, eps=mods[0].eps))
            parts = [n[:, g * og:(g + 1) * og] for g in range(self.groups)]
        else:
            z = self._lower(z)
            parts = [mods[g](z[:, g * og:(g + 1) * og]) for g in range(self.groups)]
        parts = [self.prelus[g](t) for g, t in enumerate(parts)]
        return parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)


# =========================================================================================== B-spline
class KANConvNDLayer(_HipLayer):
    def __init__(self, conv_class, norm_class, input_dim, output_dim, spline_order, kernel_size,
                 groups=1, padding=0, stride=1, dilation=1,
                 ndim: int = 2, grid_size=5, base_activation=nn.GELU, grid_range=[-1, 1], dropout=0.0,
                 **norm_kwargs):
        super().__init__()
        _need_conv2d(conv_class, ndim, allow_3d=True)
        self.input_dim, self.output_dim = input_dim, output_dim
        self.spline_order, self.kernel_size = spline_order, kernel_size
        self.padding, self.stride, self.dilation, self.groups, self.ndim = padding, stride, dilation, groups, ndim
        self.grid_size = grid_size
        self.base_activation = base_activation() if base_activation is not None else nn.Identity()
        self.grid_range = grid_range
        self.norm_kwargs = norm_kwargs
        self.dropout = _dropout2d(dropout, ndim)
        _check_groups(groups, input_dim, output_dim)
        self.input_dim_group, self.output_dim_group = input_dim // groups, output_dim // groups

        cg, og = self.input_dim_group, self.output_dim_group
        self.base_conv = nn.ModuleList([conv_class(cg, og, kernel_size, stride, padding, dilation, groups=1, bias=False)
                                        for _ in range(groups)])
        self.spline_conv = nn.ModuleList([conv_class((grid_size + spline_order) * cg, og, kernel_size, stride, padding, dilation,
                                                     groups=1, bias=False) for _ in range(groups)])
        self.layer_norm = nn.ModuleList([norm_class(og, **_filter_norm_kwargs(norm_class, norm_kwargs)) for _ in range(groups)])
        self.prelus = nn.ModuleList([nn.PReLU() for _ in range(groups)])

        h = (self.grid_range[1] - self.grid_range[0]) / grid_size
        # plain attribute, not a buffer: absent from state_dict exactly as in kan_layers.py:184-190
        self.grid = torch.linspace(self.grid_range[0] - h * spline_order, self.grid_range[1] + h * spline_order,
                                   grid_size + 2 * spline_order + 1, dtype=torch.float32)
        for conv in self.base_conv:
            nn.init.kaiming_uniform_(conv.weight, nonlinearity='linear')
        for conv in self.spline_conv:
            nn.init.kaiming_uniform_(conv.weight, nonlinearity='linear')
        self._act_code = _act_code(self.base_activation, host_ok=True)

    def _basis_kw(self):
        return dict(kind=L.BASIS_BSPLINE, n_basis=self.grid_size + self.spline_order, order=self.spline_order,
                    act=self._act_code, p0=0.0, p1=0.0, table=tuple(float(v) for v in self.grid.tolist()))

    def conv_spec(self) -> ops.ConvSpec:
        return self._spec(**self._basis_kw())

    def _plane_windows(self):
        """The library holds at most KAN_MAX_PLANES = 16 planes per channel in one launch; the reference takes any grid_size
        (kan_layers.py:117-131).  B-spline basis j is a function of knots j .. j + order + 1 alone, so bases [j0, j1) of this layer ARE the
        bases of a B-spline layer built on knots[j0 : j1 + order + 1] -- and the conv stage is linear in the planes.  A layer of more than
        16 planes therefore runs one launch set per window of <= 16 planes (the first carries the base branch) and sums the results:
        [(spec, j0, j1, with_base), ...]."""
        n, S, knots = self.grid_size + self.spline_order, self.spline_order, [float(v) for v in self.grid.tolist()]
        out, j0, first = [], 0, True
        while j0 < n:
            has_base = first and self._act_code != L.ACT_NONE
            j1 = min(n, j0 + L.KAN_MAX_PLANES - (1 if has_base else 0))
            out.append((self._spec(kind=L.BASIS_BSPLINE, n_basis=j1 - j0, order=S, act=self._act_code if has_base else L.ACT_NONE, p0=0.0, p1=0.0,
                                   table=tuple(knots[j0:j1 + S + 1])), j0, j1, has_base))
            j0, first = j1, False
        return out

    def _windowed_stage(self, xa, xb, wb, ws):
        """Conv stage of a layer with more than 16 planes per channel: sum over plane windows (differentiable: autograd routes the
        gradient of each weight slice back into spline_conv[g].weight)."""
        n = self.grid_size + self.spline_order
        z = None
        for spec, j0, j1, has_base in self._plane_windows():
            wsl = [w.reshape(w.shape[0], w.shape[1] // n, n, *w.shape[2:])[:, :, j0:j1].reshape(w.shape[0], -1, *w.shape[2:]) for w in ws]
            part = ops.kan_conv(spec, xa if has_base else (xb if xb is not None else xa), None if (not has_base or xb is None) else xb,
                                wb if has_base else [], wsl)
            z = part if z is None else z + part
        return z

    def _forward3d(self, x):
        xa, xb = self._base_input(x)
        z = conv3d_stage(self._basis_kw(), self.kernel_size, self.stride, self.padding, self.dilation, self.groups, xa, xb,
                         [m.weight for m in self.base_conv], [m.weight for m in self.spline_conv])
        y = _norm3d(self.layer_norm, self.prelus, z, self.output_dim_group)
        return self.dropout(y) if self.dropout is not None else y

    def forward(self, x, pool=False):
        """`pool=True` (not part of the reference signature; used by models/kan_vgg.py for a layer that is followed by
        MaxPool2d(2, 2)) returns max_pool2d(layer(x), 2, 2) with the pooling done inside the InstanceNorm+PReLU kernels;
        `pool=(k, s)` likewise for a general MaxPool2d(k, s) without padding (models/kan_alexnet.py: (3, 2))."""
        if self.ndim == 3:
            if pool:
                raise NotImplementedError("pool=True is a 2-D fusion")
            return self._forward3d(x)
        spec = self.conv_spec()
        x = self._lift(x)
        wb, ws = self._w(self.base_conv), self._w(self.spline_conv)
        prelus = [m.weight for m in self.prelus]
        xa, xb = self._base_input(x)
        if spec.n_basis + int(spec.has_base) > L.KAN_MAX_PLANES:             # more planes than one launch holds: plane windows, un-fused tail
            y = self._norm_prelu(self._windowed_stage(xa, xb, wb, ws))
            if self.dropout is not None:
                y = self.dropout(y)
            return _unfused_pool(y, pool) if pool else y
        if xb is None and _fusable_instnorm(self.layer_norm) and all(p.numel() == 1 for p in prelus):
            gam, bet = self._norm_affine(self.layer_norm)
            if pool and self.ndim == 2 and self.dropout is None:
                ho, wo = spec.out_hw(x.shape[2], x.shape[3])
                if pool is not True or (ho % 2 == 0 and wo % 2 == 0):
                    return ops.kan_conv_in_prelu(spec, x, wb, ws, gam, bet, prelus, eps=self.layer_norm[0].eps, pool=pool)
            y = self._lower(ops.kan_conv_in_prelu(spec, x, wb, ws, gam, bet, prelus, eps=self.layer_norm[0].eps))
        else:
            # other norm classes (e.g. BatchNorm2d) or a host-applied activation: HIP conv stage, then the un-fused tail
            y = self._norm_prelu(ops.kan_conv(spec, xa, xb, wb, ws))
        if self.dropout is not None:
            y = self.dropout(y)
        return _unfused_pool(y, pool) if pool else y


class KANConv3DLayer(KANConvNDLayer):
    def __init__(self, input_dim, output_dim, kernel_size, spline_order=3, groups=1, padding=0, stride=1, dilation=1,
                 grid_size=5, base_activation=nn.GELU, grid_range=[-1, 1], dropout=0.0, norm_layer=nn.InstanceNorm3d,
                 **norm_kwargs):
        super().__init__(nn.Conv3d, norm_layer, input_dim, output_dim, spline_order, kernel_size,
                         groups=groups, padding=padding, stride=stride, dilation=dilation, ndim=3,
                         grid_size=grid_size, base_activation=base_activation, grid_range=grid_range, dropout=dropout, **norm_kwargs)


class KANConv1DLayer(KANConvNDLayer):
    def __init__(self, input_dim, output_dim, kernel_size, spline_order=3, groups=1, padding=0, stride=1, dilation=1,
                 grid_size=5, base_activation=nn.GELU, grid_range=[-1, 1], dropout=0.0, norm_layer=nn.InstanceNorm1d,
                 **norm_kwargs):
        super().__init__(nn.Conv1d, norm_layer, input_dim, output_dim, spline_order, kernel_size,
                         groups=groups, padding=padding, stride=stride, dilation=dilation, ndim=1,
                         grid_size=grid_size, base_activation=base_activation, grid_range=grid_range, dropout=dropout,
                         **norm_kwargs)


class KANConv2DLayer(KANConvNDLayer):
    def __init__(self, input_dim, output_dim, kernel_size, spline_order=3, groups=1, padding=0, stride=1, dilation=1,
                 grid_size=5, base_activation=nn.GELU, grid_range=[-1, 1], dropout=0.0, norm_layer=nn.InstanceNorm2d,
                 **norm_kwargs):
        super().__init__(nn.Conv2d, norm_layer, input_dim, output_dim, spline_order, kernel_size,
                         groups=groups, padding=padding, stride=stride, dilation=dilation, ndim=2,
                         grid_size=grid_size, base_activation=base_activation, grid_range=grid_range, dropout=dropout,
                         **norm_kwargs)


# =========================================================================================== FastKAN
class RadialBasisFunction(nn.Module):
    """Parameter holder mirroring utils/utils.py:19-33 (state_dict key ``rbf.grid``)."""

    def __init__(self, grid_min: float = -2., grid_max: float = 2., num_grids: int = 8, denominator: Optional[float] = None):
        super().__init__()
        self.grid = nn.Parameter(torch.linspace(grid_min, grid_max, num_grids), requires_grad=False)
        self.denominator = denominator or (grid_max - grid_min) / (num_grids - 1)


class FastKANConvNDLayer(_HipLayer):
    def __init__(self, conv_class, norm_class, input_dim, output_dim, kernel_size,
                 groups=1, padding=0, stride=1, dilation=1,
                 ndim: int = 2, grid_size=8, base_activation=nn.SiLU, grid_range=[-2, 2], dropout=0.0, **norm_kwargs):
        super().__init__()
        _need_conv2d(conv_class, ndim, allow_3d=True)
        self.input_dim, self.output_dim, self.kernel_size = input_dim, output_dim, kernel_size
        self.padding, self.stride, self.dilation, self.groups, self.ndim = padding, stride, dilation, groups, ndim
        self.grid_size = grid_size
        self.base_activation = base_activation() if base_activation is not None else nn.Identity()
        self.grid_range = grid_range
        self.norm_kwargs = norm_kwargs
        _check_groups(groups, input_dim, output_dim)
        cg, og = input_dim // groups, output_dim // groups
        self.base_conv = nn.ModuleList([conv_class(cg, og, kernel_size, stride, padding, dilation, groups=1, bias=False)
                                        for _ in range(groups)])
        self.spline_conv = nn.ModuleList([conv_class(grid_size * cg, og, kernel_size, stride, padding, dilation, groups=1, bias=False)
                                          for _ in range(groups)])
        self.layer_norm = nn.ModuleList([norm_class(cg, **_filter_norm_kwargs(norm_class, norm_kwargs)) for _ in range(groups)])
        self.rbf = RadialBasisFunction(grid_range[0], grid_range[1], grid_size)
        self.dropout = _dropout2d(dropout, ndim)
        for conv in self.base_conv:
            nn.init.kaiming_uniform_(conv.weight, nonlinearity='linear')
        for conv in self.spline_conv:
            nn.init.kaiming_uniform_(conv.weight, nonlinearity='linear')
        self._act_code = _act_code(self.base_activation, host_ok=True)
        self._centres = tuple(float(v) for v in self.rbf.grid.detach().tolist())

    def _basis_kw(self):
        return dict(kind=L.BASIS_RBF, n_basis=self.grid_size, order=0, act=self._act_code, p0=float(self.rbf.denominator), p1=0.0,
                    table=self._centres)

    def conv_spec(self) -> ops.ConvSpec:
        return self._spec(**self._basis_kw())

    def _forward3d(self, x):
        xs = self.dropout(x) if self.dropout is not None else x
        cg = self.input_dim // self.groups
        if all(type(m) is nn.InstanceNorm3d and not m.track_running_stats for m in self.layer_norm):
            B, C, D, H, W = xs.shape
            gam = torch.cat([m.weight for m in self.layer_norm]) if self.layer_norm[0].affine else None
            bet = torch.cat([m.bias for m in self.layer_norm]) if self.layer_norm[0].affine else None
            xn = ops.instance_norm(xs.reshape(B, C, D * H, W), gam, bet, eps=self.layer_norm[0].eps).view(B, C, D, H, W)
        else:
            xn = torch.cat([self.layer_norm[g](xs[:, g * cg:(g + 1) * cg]) for g in range(self.groups)], dim=1)
        return conv3d_stage(self._basis_kw(), self.kernel_size, self.stride, self.padding, self.dilation, self.groups, self._base_input(x)[0], xn,
                            [m.weight for m in self.base_conv], [m.weight for m in self.spline_conv])

    def forward(self, x):
        if self.ndim == 3:
            return self._forward3d(x)
        # fast_kan_layers.py:100-111: the base branch sees raw x; the RBFs see norm(dropout(x))
        xs = self.dropout(x) if self.dropout is not None else x
        cg = self.input_dim // self.groups
        if _fusable_instnorm(self.layer_norm):
            if self.layer_norm[0].affine:
                gam = torch.cat([m.weight for m in self.layer_norm])
                bet = torch.cat([m.bias for m in self.layer_norm])
            else:
                gam = bet = None
            xn = self._lift(xs) if xs.dim() == 3 else xs
            xn = ops.instance_norm(xn.contiguous(), gam, bet, eps=self.layer_norm[0].eps)
        else:
            xn = self._lift(torch.cat([self.layer_norm[g](xs[:, g * cg:(g + 1) * cg]) for g in range(self.groups)], dim=1))
        return self._lower(ops.kan_conv(self.conv_spec(), self._lift(self._base_input(x)[0]), xn, self._w(self.base_conv), self._w(self.spline_conv)))


class FastKANConv1DLayer(FastKANConvNDLayer):
    def __init__(self, input_dim, output_dim, kernel_size, groups=1, padding=0, stride=1, dilation=1,
                 grid_size=8, base_activation=nn.SiLU, grid_range=[-2, 2], dropout=0.0,
                 norm_layer=nn.InstanceNorm1d, **norm_kwargs):
        super().__init__(nn.Conv1d, norm_layer, input_dim, output_dim, kernel_size,
                         groups=groups, padding=padding, stride=stride, dilation=dilation, ndim=1,
                         grid_size=grid_size, base_activation=base_activation, grid_range=grid_range,
                         dropout=dropout, **norm_kwargs)


class FastKANConv3DLayer(FastKANConvNDLayer):
    def __init__(self, input_dim, output_dim, kernel_size, groups=1, padding=0, stride=1, dilation=1,
                 grid_size=8, base_activation=nn.SiLU, grid_range=[-2, 2], dropout=0.0,
                 norm_layer=nn.InstanceNorm3d, **norm_kwargs):
        super().__init__(nn.Conv3d, norm_layer, input_dim, output_dim, kernel_size,
                         groups=groups, padding=padding, stride=stride, dilation=dilation, ndim=3,
                         grid_size=grid_size, base_activation=base_activation, grid_range=grid_range,
                         dropout=dropout, **norm_kwargs)


class FastKANConv2DLayer(FastKANConvNDLayer):
    def __init__(self, input_dim, output_dim, kernel_size, groups=1, padding=0, stride=1, dilation=1,
                 grid_size=8, base_activation=nn.SiLU, grid_range=[-2, 2], dropout=0.0,
                 norm_layer=nn.InstanceNorm2d, **norm_kwargs):
        super().__init__(nn.Conv2d, norm_layer, input_dim, output_dim, kernel_size,
                         groups=groups, padding=padding, stride=stride, dilation=dilation, ndim=2,
                         grid_size=grid_size, base_activation=base_activation, grid_range=grid_range,
                         dropout=dropout, **norm_kwargs)


# =========================================================================================== ChebyKAN
class ChebyKANConvNDLayer(_HipLayer):
    def __init__(self, conv_class, norm_layer, input_dim, output_dim, degree, kernel_size,
                 groups=1, padding=0, stride=1, dilation=1, ndim: int = 2, dropout=0.0, **norm_kwargs):
        super().__init__()
        _need_conv2d(conv_class, ndim, allow_3d=True)
        self.input_dim, self.output_dim, self.degree, self.kernel_size = input_dim, output_dim, degree, kernel_size
        self.padding, self.stride, self.dilation, self.groups, self.ndim = padding, stride, dilation, groups, ndim
        self.norm_kwargs = norm_kwargs
        self.epsilon = 1e-7
        self.dropout = _dropout2d(dropout, ndim)
        _check_groups(groups, input_dim, output_dim)
        og = output_dim // groups
        self.layer_norm = nn.ModuleList([norm_layer(og, **_filter_norm_kwargs(norm_layer, norm_kwargs)) for _ in range(groups)])
        self.poly_conv = nn.ModuleList([conv_class((degree + 1) * input_dim // groups, og, kernel_size, stride, padding, dilation,
                                                   groups=1, bias=False) for _ in range(groups)])
        self.register_buffer("arange", torch.arange(0, degree + 1, 1).view(1, 1, -1, *([1] * ndim)))    # cheby_kan_layers.py:85-86
        for conv in self.poly_conv:
            # cheby_kan_layers.py:88-90 (normal_ first, then overwritten; `**` requires an int kernel_size, as there)
            nn.init.normal_(conv.weight, mean=0.0, std=1 / (input_dim * (degree + 1) * kernel_size ** ndim))
            nn.init.kaiming_normal_(conv.weight, mode='fan_in', nonlinearity='relu')

    def _basis_kw(self):
        lo = float(np.float32(-1 + self.epsilon))       # torch.clamp casts its Python-float bounds to fp32
        hi = float(np.float32(1 - self.epsilon))
        return dict(kind=L.BASIS_CHEBY, n_basis=self.degree + 1, order=0, act=L.ACT_NONE, p0=lo, p1=hi, table=())

    def conv_spec(self) -> ops.ConvSpec:
        return self._spec(**self._basis_kw())

    def forward(self, x, pool=False):
        """`pool` (not part of the reference signature; models/kan_alexnet.py passes (3, 2) for a layer followed by MaxPool2d(3, 2)): True or a
        (kernel, stride) pair -- returns max_pool2d(layer(x), kernel, stride) with the pooling done inside the InstanceNorm kernels."""
        if self.ndim == 3:
            if pool:
                raise NotImplementedError("pool is a 2-D fusion")
            z = conv3d_stage(self._basis_kw(), self.kernel_size, self.stride, self.padding, self.dilation, self.groups, x, None, [],
                             [m.weight for m in self.poly_conv])
            y = _norm3d(self.layer_norm, None, z, self.output_dim // self.groups)
            return self.dropout(y) if self.dropout is not None else y
        spec = self.conv_spec()
        x = self._lift(x)
        wp = self._w(self.poly_conv)
        if pool and self.ndim == 2 and self.dropout is None and _fusable_instnorm(self.layer_norm):
            gam, bet = self._norm_affine(self.layer_norm)
            return ops.kan_conv_in_prelu(spec, x, [], wp, gam, bet, None, eps=self.layer_norm[0].eps, pool=pool)
        if _fusable_instnorm(self.layer_norm):
            gam, bet = self._norm_affine(self.layer_norm)
            y = self._lower(ops.kan_conv_in_prelu(spec, x, [], wp, gam, bet, None, eps=self.layer_norm[0].eps))
        else:
            z = self._lower(ops.kan_conv(spec, x, None, [], wp))
            og = self.output_dim // self.groups
            y = torch.cat([self.layer_norm[g](z[:, g * og:(g + 1) * og]) for g in range(self.groups)], dim=1)
        if self.dropout is not None:
            y = self.dropout(y)
        return _unfused_pool(y, pool) if pool else y


class ChebyKANConv1DLayer(ChebyKANConvNDLayer):
    def __init__(self, input_dim, output_dim, kernel_size, degree=3, groups=1, padding=0, stride=1, dilation=1,
                 dropout=0.0, norm_layer=nn.InstanceNorm1d, **norm_kwargs):
        super().__init__(nn.Conv1d, norm_layer, input_dim, output_dim, degree, kernel_size,
                         groups=groups, padding=padding, stride=stride, dilation=dilation, ndim=1, dropout=dropout, **norm_kwargs)


class ChebyKANConv3DLayer(ChebyKANConvNDLayer):
    def __init__(self, input_dim, output_dim, kernel_size, degree=3, groups=1, padding=0, stride=1, dilation=1,
                 dropout=0.0, norm_layer=nn.InstanceNorm3d, **norm_kwargs):
        super().__init__(nn.Conv3d, norm_layer, input_dim, output_dim, degree, kernel_size,
                         groups=groups, padding=padding, stride=stride, dilation=dilation, ndim=3, dropout=dropout, **norm_kwargs)


class ChebyKANConv2DLayer(ChebyKANConvNDLayer):
    def __init__(self, input_dim, output_dim, kernel_size, degree=3, groups=1, padding=0, stride=1, dilation=1,
                 dropout=0.0, norm_layer=nn.InstanceNorm2d, **norm_kwargs):
        super().__init__(nn.Conv2d, norm_layer, input_dim, output_dim, degree, kernel_size,
                         groups=groups, padding=padding, stride=stride, dilation=dilation, ndim=2, dropout=dropout, **norm_kwargs)


__all__ = ["KANConvNDLayer", "KANConv2DLayer", "KANConv1DLayer", "KANConv3DLayer", "FastKANConv3DLayer", "ChebyKANConv3DLayer", "FastKANConvNDLayer", "FastKANConv2DLayer", "FastKANConv1DLayer",
           "ChebyKANConvNDLayer", "ChebyKANConv2DLayer", "ChebyKANConv1DLayer", "RadialBasisFunction"]
_ = F
