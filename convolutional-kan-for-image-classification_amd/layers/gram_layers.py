"""GRAM-KAN conv layers -- SURVEY.md section 8(f) (widening past rank 3: a basis with trainable, layer-global parameters).

  reference class (layers/gram_kan_layers.py)      this file
  :85-200   GRAMKANConvNDLayer                     GRAMKANConvNDLayer
  :212-219  GRAMKANConv2DLayer                     GRAMKANConv2DLayer

    y = act(norm(conv(act(x), W_base) + conv(act(P(tanh x)), poly_weights[g])))                    (:172-189)

Gram polynomials P_0 = 1, P_1 = t, P_k = t P_{k-1} - beta(k-1, k) P_{k-2} with
beta(n, m) = (m+n)(m-n) n^2 / (m^2 / (4 n^2 - 1)) * beta_weights[n]  (:150-170), planes concatenated PLANE-major (k*C + c)
and passed through the layer's activation.  ``beta_weights`` is trainable: the coefficients are formed with torch ops on the
device (no host read-back), handed to the fused conv stage as a device table (KAN_BASIS_GRAM), and their gradient comes from
the weight-gradient kernel run on the coefficient-derivative planes (ops._KanConvPhased) -- autograd carries it on to
``beta_weights`` through the constant factors.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops
from .conv_layers import _HipLayer, _act_code, _check_groups, _filter_norm_kwargs, _fusable_instnorm, _need_conv2d


class GRAMKANConvNDLayer(_HipLayer):
    def __init__(self, conv_class, norm_class, conv_w_fun, input_dim, output_dim, degree, kernel_size, base_activation=nn.SiLU,
                 groups=1, padding=0, stride=1, dilation=1, dropout: float = 0.0, ndim: int = 2, **norm_kwargs):
        super().__init__()
        ndim = int(ndim)
        if conv_class is not nn.Conv2d or ndim != 2:
            raise NotImplementedError("GRAM-KAN is built for 2-D only")
        self.input_dim, self.output_dim, self.degree, self.kernel_size = input_dim, output_dim, degree, kernel_size
        self.padding, self.stride, self.dilation, self.groups = padding, stride, dilation, groups
        self.base_activation = base_activation() if base_activation is not None else nn.Identity()
        self.conv_w_fun, self.ndim, self.norm_kwargs, self.p_dropout = conv_w_fun, ndim, norm_kwargs, dropout
        self.dropout = None
        if dropout > 0:
            raise NotImplementedError("GRAM-KAN applies dropout between tanh and the polynomial (gram_kan_layers.py:178-179); the "
                                      "fused conv stage never materialises that tensor -- use dropout=0")
        _check_groups(groups, input_dim, output_dim)
        if degree < 1 or degree + 2 > L.KAN_MAX_PLANES:
            raise NotImplementedError(f"GRAM-KAN on the HIP path needs 1 <= degree <= {L.KAN_MAX_PLANES - 2}")
        if not isinstance(kernel_size, int):
            raise TypeError("GRAM-KAN takes an int kernel_size (gram_kan_layers.py:132-133,146)")
        cg, og = input_dim // groups, output_dim // groups
        self.base_conv = nn.ModuleList([conv_class(cg, og, kernel_size, stride, padding, dilation, groups=1, bias=False)
                                        for _ in range(groups)])
        self.layer_norm = nn.ModuleList([norm_class(og, **_filter_norm_kwargs(norm_class, norm_kwargs)) for _ in range(groups)])
        self.poly_weights = nn.Parameter(torch.randn(groups, og, cg * (degree + 1), *([kernel_size] * ndim)))
        self.beta_weights = nn.Parameter(torch.zeros(degree + 1, dtype=torch.float32))
        for conv in self.base_conv:
            nn.init.kaiming_uniform_(conv.weight, nonlinearity='linear')
        nn.init.kaiming_uniform_(self.poly_weights, nonlinearity='linear')
        nn.init.normal_(self.beta_weights, mean=0.0, std=1.0 / ((kernel_size ** ndim) * input_dim * (degree + 1.0)))
        # c_k = beta(k-1, k) = factor[k] * beta_weights[k-1]; the factor is the reference's Python-float expression (:150-153)
        fac = [0.0, 0.0] + [((2 * i - 1) * 1 * (i - 1) ** 2) / (i ** 2 / (4.0 * (i - 1) ** 2 - 1.0)) for i in range(2, degree + 1)]
        self.register_buffer("_beta_factor", torch.tensor(fac, dtype=torch.float32), persistent=False)

    def conv_spec(self) -> ops.ConvSpec:
        return self._spec(kind=L.BASIS_GRAM, n_basis=self.degree + 1, order=0, act=_act_code(self.base_activation), p0=0.0, p1=0.0,
                          table=())

    def forward(self, x):
        G, n = self.groups, self.degree + 1
        og, cg, k = self.output_dim // G, self.input_dim // G, self.kernel_size
        coef = self._beta_factor * torch.cat([self.beta_weights.new_zeros(1), self.beta_weights[:-1]])      # coef[k] = c_k
        # plane-major (k*C + c) -> the kernels' channel-major (c*n + k) order; autograd carries the gradient back
        ws = [self.poly_weights[g].view(og, n, cg, k, k).transpose(1, 2).reshape(og, cg * n, k, k) for g in range(G)]
        z = ops.kan_conv_phased(self.conv_spec(), x, coef, [m.weight for m in self.base_conv], ws)
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


class GRAMKANConv2DLayer(GRAMKANConvNDLayer):
    def __init__(self, input_dim, output_dim, kernel_size, degree=3, groups=1, padding=0, stride=1, dilation=1, dropout: float = 0.0,
                 norm_layer=nn.InstanceNorm2d, **norm_kwargs):
        super().__init__(nn.Conv2d, norm_layer, torch.nn.functional.conv2d, input_dim, output_dim, degree, kernel_size, groups=groups,
                         padding=padding, stride=stride, dilation=dilation, ndim=2, dropout=dropout, **norm_kwargs)
