"""MLP KAN layer (B-spline) on the same HIP conv stage -- SURVEY.md section 8(f), "next" rank 2.

  reference                                         this file
  layers/kan_layers.py:8-114     KANLayer            KANLayer   (same ctor, parameter names and state_dict keys)
  models/kans.py:300-327         KAN (MLP)           KAN
  models/kans.py:481-485,556-574 mlp_kan / factory   mlp_kan, MLP_KAN_FACTORY["KAN"]

A KANLayer is the conv layer's algebra on a 1x1 image with a 1x1 kernel:
``F.linear(act(x), W_b) + F.linear(B(x).flatten, W_s.view(O, I*n))`` with the basis index minor (kan_layers.py:104-106),
which is exactly the packed channel order c*n+k of the conv kernels.  The LayerNorm + PReLU epilogue (kan_layers.py:109-110)
is a [B, O] elementwise tail and stays on torch ops.
"""
from typing import List, Type

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L
from .. import ops
from .conv_layers import _act_code, _host_applied


class KANLayer(nn.Module):
    def __init__(self, input_features, output_features, grid_size=5, spline_order=3, base_activation=nn.GELU, grid_range=[-1, 1]):
        super().__init__()
        self.input_features, self.output_features = input_features, output_features
        self.grid_size, self.spline_order = grid_size, spline_order
        self.base_activation = base_activation() if base_activation is not None else nn.Identity()
        self.grid_range = grid_range
        self.base_weight = nn.Parameter(torch.randn(output_features, input_features))
        self.spline_weight = nn.Parameter(torch.randn(output_features, input_features, grid_size + spline_order))
        self.layer_norm = nn.LayerNorm(output_features)
        self.prelu = nn.PReLU()
        h = (self.grid_range[1] - self.grid_range[0]) / grid_size
        # plain attribute [in, G+2S+1] as in the reference (every row is the same linspace)
        self.grid = torch.linspace(self.grid_range[0] - h * spline_order, self.grid_range[1] + h * spline_order,
                                   grid_size + 2 * spline_order + 1, dtype=torch.float32).expand(input_features, -1).contiguous()
        nn.init.kaiming_uniform_(self.base_weight, nonlinearity='linear')
        nn.init.kaiming_uniform_(self.spline_weight, nonlinearity='linear')
        self._act_code = _act_code(self.base_activation, host_ok=True)

    def conv_spec(self) -> ops.ConvSpec:
        return ops.ConvSpec(kind=L.BASIS_BSPLINE, n_basis=self.grid_size + self.spline_order, order=self.spline_order, act=self._act_code,
                            p0=0.0, p1=0.0, table=tuple(float(v) for v in self.grid[0].tolist()), kernel=(1, 1), stride=(1, 1),
                            padding=(0, 0), dilation=(1, 1), groups=1)

    def forward(self, x):
        if x.dim() != 2:
            raise ValueError("KANLayer expects [batch, features] (kan_layers.py:100-104 views the bases as [batch, in*n])")
        B, I, O = x.shape[0], self.input_features, self.output_features
        host = _host_applied(self.base_activation)            # no device functor: act(x) feeds the base branch, x the splines
        xa = self.base_activation(x) if host else x
        z = ops.kan_conv(self.conv_spec(), xa.reshape(B, I, 1, 1), x.reshape(B, I, 1, 1) if host else None, [self.base_weight.view(O, I, 1, 1)],
                         [self.spline_weight.view(O, I * (self.grid_size + self.spline_order), 1, 1)])
        return self.prelu(self.layer_norm(z.view(B, O)))


class KAN(nn.Module):
    """Stack of KANLayers with the optional dropouts of models/kans.py:300-327."""

    def __init__(self, layers_hidden, dropout: float = 0.0, grid_size=5, spline_order=3, base_activation: Type[nn.Module] = nn.GELU,
                 grid_range: List = [-1, 1], l1_decay: float = 0.0, first_dropout: bool = True, **kwargs):
        super().__init__()
        if l1_decay > 0:
            raise NotImplementedError("l1_decay > 0 wraps layers in utils.regularization.L1 in the reference; out of scope here")
        self.layers_hidden, self.grid_size, self.spline_order = layers_hidden, grid_size, spline_order
        self.base_activation, self.grid_range = base_activation, grid_range
        self.layers = nn.ModuleList([])
        if dropout > 0 and first_dropout:
            self.layers.append(nn.Dropout(p=dropout))
        self.num_layers = len(layers_hidden[:-1])
        for i, (fin, fout) in enumerate(zip(layers_hidden[:-1], layers_hidden[1:])):
            self.layers.append(KANLayer(fin, fout, grid_size=grid_size, spline_order=spline_order, base_activation=base_activation,
                                        grid_range=grid_range))
            if dropout > 0 and i != self.num_layers - 1:
                self.layers.append(nn.Dropout(p=dropout))

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
        return x


def mlp_kan(layers_hidden: List[int], dropout: float = 0.0, grid_size: int = 5, spline_order: int = 3,
            base_activation: Type[nn.Module] = nn.GELU, grid_range: List = [-1, 1], l1_decay: float = 0.0,
            first_dropout: bool = True) -> KAN:
    return KAN(layers_hidden, dropout=dropout, grid_size=grid_size, spline_order=spline_order, base_activation=base_activation,
               grid_range=grid_range, l1_decay=l1_decay, first_dropout=first_dropout)


MLP_KAN_FACTORY = {"KAN": mlp_kan}
_ = F
