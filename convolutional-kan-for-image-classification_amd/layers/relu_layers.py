"""ReLU-KAN conv layers -- SURVEY.md section 8(f), "next" rank 3 (the one family whose basis carries trainable parameters).

  reference class (layers/relu_kan_layers.py)          this file
  :41-152   ReLUConvNDLayer                            ReLUConvNDLayer
  :166-175  ReLUKANConv2DLayer                         ReLUKANConv2DLayer
  :178-187  ReLUKANConv1DLayer                         ReLUKANConv1DLayer   (lifted to 2-D like the other 1-D shims)

    y = act(norm(conv(act(x), W_base) + conv(((x - lo)_+ (hi - x)_+ r)^2, W_relukan)))          (:118-136)

with g + k planes per channel, channel index c*(g+k)+j, r = 4 g^2 / (k+1)^2, and per-channel phases ``phase_low`` /
``phase_high`` of shape (1, C/groups, g+k, 1, 1) that are trainable by default (``train_ab``) and SHARED by the groups.
The conv stage (forward, input gradient, weight gradient) is the fused HIP kernel with the phases read from device memory
(KAN_BASIS_RELU); the phase gradients come from two more runs of the weight-gradient kernel on the phase-derivative planes
(ops._KanConvPhased) -- the expanded tensor the reference materialises never exists.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops
from .conv_layers import _HipLayer, _act_code, _check_groups, _dropout2d, _filter_norm_kwargs, _fusable_instnorm, _need_conv2d


class ReLUConvNDLayer(_HipLayer):
    def __init__(self, conv_class, norm_class, conv_w_fun, input_dim, output_dim, kernel_size, g: int = 5, k: int = 3,
                 base_activation=nn.SiLU, groups=1, padding=0, stride=1, dilation=1, dropout: float = 0.0, ndim: int = 2,
                 train_ab: bool = True, **norm_kwargs):
        super().__init__()
        ndim = int(ndim)                                         # the reference's default is the float 2.
        _need_conv2d(conv_class, ndim)
        self.input_dim, self.output_dim, self.g, self.k = input_dim, output_dim, g, k
        self.r = 4 * g * g / ((k + 1) * (k + 1))
        self.train_ab, self.kernel_size = train_ab, kernel_size
        self.padding, self.stride, self.dilation, self.groups = padding, stride, dilation, groups
        self.base_activation = base_activation() if base_activation is not None else nn.Identity()
        self.conv_w_fun, self.ndim, self.norm_kwargs, self.p_dropout = conv_w_fun, ndim, norm_kwargs, dropout
        self.dropout = _dropout2d(dropout, ndim)
        _check_groups(groups, input_dim, output_dim)
        if g + k + 1 > L.KAN_MAX_PLANES:
            raise NotImplementedError(f"g + k + 1 = {g + k + 1} planes per channel exceed KAN_MAX_PLANES = {L.KAN_MAX_PLANES}")
        cg, og = input_dim // groups, output_dim // groups
        self.base_conv = nn.ModuleList([conv_class(cg, og, kernel_size, stride, padding, dilation, groups=1, bias=False)
                                        for _ in range(groups)])
        self.relukan_conv = nn.ModuleList([conv_class((g + k) * cg, og, kernel_size, stride, padding, dilation, groups=1, bias=False)
                                           for _ in range(groups)])
        phase_low = torch.arange(-k, g) / g                      # relu_kan_layers.py:97-98
        phase_high = phase_low + (k + 1) / g
        dims = (1, cg, k + g) + (1,) * ndim
        self.phase_low = nn.Parameter(phase_low[None, :].expand(cg, -1).reshape(*dims).clone(), requires_grad=train_ab)
        self.phase_high = nn.Parameter(phase_high[None, :].expand(cg, -1).reshape(*dims).clone(), requires_grad=train_ab)
        self.layer_norm = nn.ModuleList([norm_class(og, **_filter_norm_kwargs(norm_class, norm_kwargs)) for _ in range(groups)])
        for conv in self.base_conv:
            nn.init.kaiming_uniform_(conv.weight, nonlinearity='linear')
        for conv in self.relukan_conv:
            nn.init.kaiming_uniform_(conv.weight, nonlinearity='linear')

    def conv_spec(self) -> ops.ConvSpec:
        return self._spec(kind=L.BASIS_RELU, n_basis=self.g + self.k, order=0, act=_act_code(self.base_activation, host_ok=True), p0=float(self.r),
                          p1=0.0, table=())

    def forward(self, x):
        G, n = self.groups, self.g + self.k
        cg, og = self.input_dim // G, self.output_dim // G
        if self.dropout is not None:
            x = self.dropout(x)                                  # relu_kan_layers.py:120-121: on the input, both branches see it
        phases = torch.stack([self.phase_low.reshape(cg, n), self.phase_high.reshape(cg, n)], dim=1)
        xa, xb = self._base_input(x)                              # (act(x), x) when the host applies the activation (no device functor for the module)
        z = ops.kan_conv_phased(self.conv_spec(), self._lift(xa), phases, self._w(self.base_conv), self._w(self.relukan_conv),
                                xn=self._lift(xb) if xb is not None else None)
        if _fusable_instnorm(self.layer_norm):
            gam, bet = self._norm_affine(self.layer_norm)
            y = ops.instance_norm(z, torch.cat(gam) if gam is not None else None, torch.cat(bet) if bet is not None else None,
                                  eps=self.layer_norm[0].eps)
            y = self._lower(y)
        else:
            z = self._lower(z)
            parts = []
            for gi in range(G):
                zg = z[:, gi * og:(gi + 1) * og]
                if isinstance(self.layer_norm[gi], nn.LayerNorm):
                    zg = self.layer_norm[gi](zg.reshape(zg.shape[0], -1)).view(zg.shape)
                else:
                    zg = self.layer_norm[gi](zg)
                parts.append(zg)
            y = torch.cat(parts, dim=1)
        return self.base_activation(y)


class ReLUKANConv2DLayer(ReLUConvNDLayer):
    def __init__(self, input_dim, output_dim, kernel_size, base_activation=nn.SiLU, g=5, k=3, train_ab=True, groups=1, padding=0,
                 stride=1, dilation=1, dropout: float = 0.0, norm_layer=nn.InstanceNorm2d, **norm_kwargs):
        super().__init__(nn.Conv2d, norm_layer, torch.nn.functional.conv2d, input_dim, output_dim, kernel_size, g=g, k=k,
                         train_ab=train_ab, base_activation=base_activation, groups=groups, padding=padding, stride=stride,
                         dilation=dilation, ndim=2, dropout=dropout, **norm_kwargs)


class ReLUKANConv1DLayer(ReLUConvNDLayer):
    def __init__(self, input_dim, output_dim, kernel_size, base_activation=nn.SiLU, g=5, k=3, train_ab=True, groups=1, padding=0,
                 stride=1, dilation=1, dropout: float = 0.0, norm_layer=nn.InstanceNorm1d, **norm_kwargs):
        # relu_kan_layers.py:183-187 does not hand `base_activation` on: the 1-D layer always runs the SiLU default
        super().__init__(nn.Conv1d, norm_layer, torch.nn.functional.conv1d, input_dim, output_dim, kernel_size, g=g, k=k,
                         train_ab=train_ab, groups=groups, padding=padding, stride=stride,
                         dilation=dilation, ndim=1, dropout=dropout, **norm_kwargs)
