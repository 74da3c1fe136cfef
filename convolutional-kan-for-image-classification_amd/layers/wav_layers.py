"""Wav-KAN conv layers (counterpart of the reference's layers/wav_kan_layers.py:113-480) on the HIP kernels.

    y = norm( wavelet_out( sum_{c, taps} Wk[o, c, tap] * psi((x_c - translation[o, c]) / scale[o, c]) ) + base_conv(SiLU(x)) )

per group (wav_kan_layers.py:430-454); dropout, when set, is applied to the input of the wavelet branch only (:433-434).  The
wavelet branch runs on the direct kernels of csrc/wavkan.inc (`ops.wav_stage`) -- every (output, input) pair has its own wavelet,
so nothing can be shared through a GEMM -- and its three reference versions ('base': one Conv(C -> 1) per output, 'fast': one
grouped Conv(O*C -> O, groups=O), 'fast_plus_one': a Conv(N+1)D over the channel axis) are the same arithmetic over differently
shaped weights, so the module trees (and state_dict keys / shapes) mirror the reference and the kernels see one [O, C, kh, kw] view.
The base conv and the 1x1 `wavelet_out` conv run on the MFMA conv kernels as two-plane launches (the activation plane + a constant
plane with zero weights); a plain InstanceNorm runs on the InstanceNorm kernel.  1-D ([B, C, L]), 2-D and 3-D layers (3-D: the depth
axis is walked tap by tap around the 2-D kernels, as the conv families' 3-D shims do).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L
from .. import ops

from .conv_layers import _HipLayer, _check_groups, _dropout2d, _filter_norm_kwargs, _fusable_instnorm, _norm3d, conv3d_stage

WAVELET_TYPES = ('mexican_hat', 'morlet', 'dog', 'meyer', 'shannon')


def _tup(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


class WaveletConvND(nn.Module):
    """Parameter holder of one group's wavelet branch, 'base' version (wav_kan_layers.py:113-217): `scale`, `translation`
    [1, O, C, 1(, 1)], one Conv(C -> 1) per output in `wavelet_weights`, the 1x1 `wavelet_out`.  forward() is the HIP path."""

    def __init__(self, conv_class, input_dim, output_dim, kernel_size, padding=0, stride=1, dilation=1, ndim: int = 2,
                 wavelet_type='mexican_hat'):
        super().__init__()
        self._params(input_dim, output_dim, ndim, wavelet_type)
        self.wavelet_weights = nn.ModuleList([conv_class(input_dim, 1, kernel_size, stride, padding, dilation, groups=1, bias=False)
                                              for _ in range(output_dim)])
        self._geometry(self.wavelet_weights[0])
        self.wavelet_out = conv_class(output_dim, output_dim, 1, 1, 0, dilation, groups=1, bias=False)
        for conv in self.wavelet_weights:
            nn.init.kaiming_uniform_(conv.weight, nonlinearity='linear')
        nn.init.kaiming_uniform_(self.wavelet_out.weight, nonlinearity='linear')

    def _params(self, input_dim, output_dim, ndim, wavelet_type):
        shapes = (1, output_dim, input_dim) + (1,) * ndim
        self.scale = nn.Parameter(torch.ones(*shapes))
        self.translation = nn.Parameter(torch.zeros(*shapes))
        self.ndim, self.wavelet_type, self.input_dim, self.output_dim = ndim, wavelet_type, input_dim, output_dim

    def _geometry(self, conv, skip: int = 0):
        """kernel / stride / padding / dilation of the spatial axes as pairs ((1, k) etc. for 1-D layers)."""
        ks, st, pd, dl = (tuple(getattr(conv, a))[skip:] for a in ("kernel_size", "stride", "padding", "dilation"))
        if len(ks) == 1:
            ks, st, pd, dl = (1, ks[0]), (1, st[0]), (0, pd[0]), (1, dl[0])
        self._depth = None
        if len(ks) == 3:                                   # 3-D: the depth axis is walked tap by tap around the 2-D kernels (stage3d)
            self._depth = (ks[0], st[0], pd[0], dl[0])
            ks, st, pd, dl = ks[1:], st[1:], pd[1:], dl[1:]
        self._ks, self._st, self._pd, self._dl = ks, st, pd, dl

    def weight_view(self) -> torch.Tensor:
        """[O, C, kh, kw] view of the wavelet weights."""
        w = torch.cat([m.weight for m in self.wavelet_weights], dim=0)                     # O x [1, C, k(, k)]
        return w.unsqueeze(2) if self.ndim == 1 else w

    def stage3d(self, x5: torch.Tensor) -> torch.Tensor:
        """3-D layers: the wavelet sum is a sum over the kd depth taps of 2-D ones -- output slice `do` takes, for tap `td`, the 2-D stage of
        input slice `do*sd - pd + td*dd` with the weights W[:, :, td], and nothing where that slice lies in the depth padding (the zero padding
        of the wavelet values); each tap is one launch over all (image, valid slice) pairs, as `conv3d_stage` does for the conv families."""
        kd, sd, pd, dd = self._depth
        B, Cn, D, H, W = x5.shape
        Do = (D + 2 * pd - dd * (kd - 1) - 1) // sd + 1
        if Do <= 0:
            raise L.KanConvError(f"empty output depth for input depth {D}")
        o, c = self.output_dim, self.input_dim
        w5, sc, tr = self.weight_view(), self.scale.view(o, c), self.translation.view(o, c)
        u = None
        for td in range(kd):
            dos = [q for q in range(Do) if 0 <= q * sd - pd + td * dd < D]
            if not dos:
                continue
            d0, n = dos[0] * sd - pd + td * dd, len(dos)
            xs = x5[:, :, d0:d0 + (n - 1) * sd + 1:sd].permute(0, 2, 1, 3, 4).reshape(B * n, Cn, H, W)
            u2 = ops.wav_stage(L.WAVELETS[self.wavelet_type], xs.contiguous(), sc, tr, w5[:, :, td], self._st, self._pd, self._dl)
            u2 = F.pad(u2.view(B, n, *u2.shape[1:]).permute(0, 2, 1, 3, 4), (0, 0, 0, 0, dos[0], Do - dos[0] - n))
            u = u2 if u is None else u + u2
        if u is None:                                          # every tap of every output slice lies in the depth padding
            ho = (H + 2 * self._pd[0] - self._dl[0] * (self._ks[0] - 1) - 1) // self._st[0] + 1
            wo = (W + 2 * self._pd[1] - self._dl[1] * (self._ks[1] - 1) - 1) // self._st[1] + 1
            u = x5.new_zeros((B, o, Do, ho, wo)) + 0.0 * (x5.sum() + w5.sum() + sc.sum() + tr.sum())
        return u

    def stage(self, x4: torch.Tensor) -> torch.Tensor:
        """The wavelet sum u (before `wavelet_out`) of a [B, C, H, W] input (1-D layers: H = 1)."""
        o, c = self.output_dim, self.input_dim
        return ops.wav_stage(L.WAVELETS[self.wavelet_type], x4, self.scale.view(o, c), self.translation.view(o, c), self.weight_view(),
                             self._st, self._pd, self._dl)


class WaveletConvNDFast(WaveletConvND):
    """'fast' version (wav_kan_layers.py:285-338): one grouped Conv(O*C -> O, groups=O), weight [O, C, k(, k)]."""

    def __init__(self, conv_class, input_dim, output_dim, kernel_size, padding=0, stride=1, dilation=1, ndim: int = 2,
                 wavelet_type='mexican_hat'):
        nn.Module.__init__(self)
        self._params(input_dim, output_dim, ndim, wavelet_type)
        self.wavelet_weights = conv_class(output_dim * input_dim, output_dim, kernel_size, stride, padding, dilation, groups=output_dim, bias=False)
        self._geometry(self.wavelet_weights)
        self.wavelet_out = conv_class(output_dim, output_dim, 1, 1, 0, dilation, groups=1, bias=False)
        nn.init.kaiming_uniform_(self.wavelet_weights.weight, nonlinearity='linear')
        nn.init.kaiming_uniform_(self.wavelet_out.weight, nonlinearity='linear')

    def weight_view(self):
        w = self.wavelet_weights.weight
        return w.unsqueeze(2) if self.ndim == 1 else w


class WaveletConvNDFastPlusOne(WaveletConvND):
    """'fast_plus_one' version (wav_kan_layers.py:221-282): a Conv(N+1)D whose first kernel axis spans the input channels,
    weight [O, 1, C, k(, k)]."""

    def __init__(self, conv_class, conv_class_d_plus_one, input_dim, output_dim, kernel_size, padding=0, stride=1, dilation=1, ndim: int = 2,
                 wavelet_type='mexican_hat'):
        nn.Module.__init__(self)
        assert ndim < 3, "fast_plus_one version suppoerts only 1D and 2D convs"
        self._params(input_dim, output_dim, ndim, wavelet_type)
        self.wavelet_weights = conv_class_d_plus_one(output_dim, output_dim, (input_dim,) + _tup(kernel_size, ndim), (1,) + _tup(stride, ndim),
                                                     (0,) + _tup(padding, ndim), (1,) + _tup(dilation, ndim), groups=output_dim, bias=False)
        self._geometry(self.wavelet_weights, skip=1)
        self.wavelet_out = conv_class(output_dim, output_dim, 1, 1, 0, dilation, groups=1, bias=False)
        nn.init.kaiming_uniform_(self.wavelet_weights.weight, nonlinearity='linear')
        nn.init.kaiming_uniform_(self.wavelet_out.weight, nonlinearity='linear')

    def weight_view(self):
        w = self.wavelet_weights.weight.squeeze(1)                                        # [O, C, k(, k)]
        return w.unsqueeze(2) if self.ndim == 1 else w


class WavKANConvNDLayer(_HipLayer):
    def __init__(self, conv_class, conv_class_plus1, norm_class, input_dim, output_dim, kernel_size,
                 groups=1, padding=0, stride=1, dilation=1, wav_version: str = 'base',
                 ndim: int = 2, dropout=0.0, wavelet_type='mexican_hat', **norm_kwargs):
        super().__init__()
        if ndim not in (1, 2, 3):
            raise NotImplementedError("Wav-KAN is built for 1-D, 2-D and 3-D on the HIP path")
        self.inputdim, self.outdim = input_dim, output_dim                # (attribute names as in wav_kan_layers.py:345-346)
        self.kernel_size, self.padding, self.stride, self.dilation, self.groups, self.ndim = kernel_size, padding, stride, dilation, groups, ndim
        self.norm_kwargs = norm_kwargs
        assert wavelet_type in WAVELET_TYPES, ValueError(f"Unsupported wavelet type: {wavelet_type}")
        self.wavelet_type = wavelet_type
        self.dropout = _dropout2d(dropout, ndim)
        _check_groups(groups, input_dim, output_dim)
        cg, og = input_dim // groups, output_dim // groups
        self.output_dim_group = og
        self.base_conv = nn.ModuleList([conv_class(cg, og, kernel_size, stride, padding, dilation, groups=1, bias=False) for _ in range(groups)])
        geo = dict(stride=stride, padding=padding, dilation=dilation, ndim=ndim, wavelet_type=wavelet_type)
        if wav_version == 'base':
            self.wavelet_conv = nn.ModuleList([WaveletConvND(conv_class, cg, og, kernel_size, **geo) for _ in range(groups)])
        elif wav_version == 'fast':
            self.wavelet_conv = nn.ModuleList([WaveletConvNDFast(conv_class, cg, og, kernel_size, **geo) for _ in range(groups)])
        elif wav_version == 'fast_plus_one':
            self.wavelet_conv = nn.ModuleList([WaveletConvNDFastPlusOne(conv_class, conv_class_plus1, cg, og, kernel_size, **geo) for _ in range(groups)])
        else:
            raise ValueError(f"unknown wav_version {wav_version!r} (base, fast, fast_plus_one)")
        self.layer_norm = nn.ModuleList([norm_class(og, **_filter_norm_kwargs(norm_class, norm_kwargs)) for _ in range(groups)])
        self.base_activation = nn.SiLU()
        # zero weights of the constant plane the two-plane conv launches carry (never trained, not in state_dict)
        self.register_buffer("_zero_base", torch.zeros(groups, og, cg, *((1,) if ndim == 1 else ()), *self.base_conv[0].weight.shape[2:]), persistent=False)
        self.register_buffer("_zero_out", torch.zeros(groups, og, og, 1, 1), persistent=False)

    def _const_plane_spec(self, act: int, kernel, stride, padding, dilation) -> ops.ConvSpec:
        """act(x) conv W + (constant plane) conv 0: a plain convolution on the KAN conv kernels."""
        return ops.ConvSpec(kind=L.BASIS_POLY, n_basis=1, order=0, act=act, p0=0.0, p1=0.0, table=(1.0, 0.0, 0.0),
                            kernel=kernel, stride=stride, padding=padding, dilation=dilation, groups=self.groups)

    def conv_spec(self) -> ops.ConvSpec:
        """Spec of the base-conv launch (geometry queries: out_hw)."""
        wc0 = self.wavelet_conv[0]
        return self._const_plane_spec(L.ACT_SILU, wc0._ks, wc0._st, wc0._pd, wc0._dl)

    def _forward3d(self, x):
        G, og = self.groups, self.output_dim_group
        cg = self.inputdim // G
        base = conv3d_stage(dict(kind=L.BASIS_POLY, n_basis=1, order=0, act=L.ACT_SILU, p0=0.0, p1=0.0, table=(1.0, 0.0, 0.0)), self.kernel_size,
                            self.stride, self.padding, self.dilation, G, x, None, [m.weight for m in self.base_conv], [self._zero_base[g] for g in range(G)])
        xd = self.dropout(x) if self.dropout is not None else x
        us = [self.wavelet_conv[g].stage3d(xd[:, g * cg:(g + 1) * cg]) for g in range(G)]
        u = us[0] if G == 1 else torch.cat(us, dim=1)
        B, O, Do, Ho, Wo = u.shape
        mixed = ops.kan_conv(self._const_plane_spec(L.ACT_IDENTITY, (1, 1), (1, 1), (0, 0), (1, 1)), u.reshape(B, O, Do * Ho, Wo), None,
                             [m.wavelet_out.weight.view(og, og, 1, 1) for m in self.wavelet_conv], [self._zero_out[g] for g in range(G)])
        return _norm3d(self.layer_norm, None, mixed.view(B, O, Do, Ho, Wo) + base, og)

    def forward(self, x):
        if self.ndim == 3:
            return self._forward3d(x)
        G, og = self.groups, self.output_dim_group
        cg = self.inputdim // G
        x4 = self._lift(x)
        wc0 = self.wavelet_conv[0]
        base = ops.kan_conv(self._const_plane_spec(L.ACT_SILU, wc0._ks, wc0._st, wc0._pd, wc0._dl), x4, None, self._w(self.base_conv),
                            [self._zero_base[g] for g in range(G)])
        xd = self._lift(self.dropout(x)) if self.dropout is not None else x4       # wav_kan_layers.py:433-436: dropout feeds the wavelets only
        us = [self.wavelet_conv[g].stage(xd[:, g * cg:(g + 1) * cg].contiguous()) for g in range(G)]
        u = us[0] if G == 1 else torch.cat(us, dim=1)
        w_out = [m.wavelet_out.weight.unsqueeze(2) if self.ndim == 1 else m.wavelet_out.weight for m in self.wavelet_conv]
        mixed = ops.kan_conv(self._const_plane_spec(L.ACT_IDENTITY, (1, 1), (1, 1), (0, 0), (1, 1)), u, None, w_out, [self._zero_out[g] for g in range(G)])
        z = mixed + base
        if _fusable_instnorm(self.layer_norm):
            gam = torch.cat([m.weight for m in self.layer_norm]) if self.layer_norm[0].affine else None
            bet = torch.cat([m.bias for m in self.layer_norm]) if self.layer_norm[0].affine else None
            return self._lower(ops.instance_norm(z, gam, bet, eps=self.layer_norm[0].eps))
        z = self._lower(z)
        parts = [self.layer_norm[g](z[:, g * og:(g + 1) * og]) for g in range(G)]
        return parts[0] if G == 1 else torch.cat(parts, dim=1)


class WavKANConv2DLayer(WavKANConvNDLayer):
    def __init__(self, input_dim, output_dim, kernel_size, groups=1, padding=0, stride=1, dilation=1,
                 dropout=0.0, wavelet_type='mexican_hat', norm_layer=nn.BatchNorm2d, wav_version: str = 'fast', **norm_kwargs):
        super().__init__(nn.Conv2d, nn.Conv3d, norm_layer, input_dim, output_dim, kernel_size,
                         groups=groups, padding=padding, stride=stride, dilation=dilation, ndim=2, dropout=dropout,
                         wavelet_type=wavelet_type, wav_version=wav_version, **norm_kwargs)


class WavKANConv1DLayer(WavKANConvNDLayer):
    def __init__(self, input_dim, output_dim, kernel_size, groups=1, padding=0, stride=1, dilation=1,
                 dropout=0.0, wavelet_type='mexican_hat', norm_layer=nn.BatchNorm1d, wav_version: str = 'fast', **norm_kwargs):
        super().__init__(nn.Conv1d, nn.Conv2d, norm_layer, input_dim, output_dim, kernel_size,
                         groups=groups, padding=padding, stride=stride, dilation=dilation, ndim=1, dropout=dropout,
                         wavelet_type=wavelet_type, wav_version=wav_version, **norm_kwargs)


class WavKANConv3DLayer(WavKANConvNDLayer):
    def __init__(self, input_dim, output_dim, kernel_size, groups=1, padding=0, stride=1, dilation=1,
                 dropout=0.0, wavelet_type='mexican_hat', norm_layer=nn.BatchNorm3d, wav_version: str = 'fast', **norm_kwargs):
        super().__init__(nn.Conv3d, None, norm_layer, input_dim, output_dim, kernel_size,
                         groups=groups, padding=padding, stride=stride, dilation=dilation, ndim=3, dropout=dropout,
                         wavelet_type=wavelet_type, wav_version=wav_version, **norm_kwargs)
