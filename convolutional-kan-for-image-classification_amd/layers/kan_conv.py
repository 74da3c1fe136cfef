"""Factory functions with the reference's signatures (layers/kan_conv.py:27-69, 197-276, 726-745).

``CONV_KAN_FACTORY[name](in_planes, out_planes, kernel_size=..., ...)`` is the drop-in boundary the
reference's models use (models/kan_vgg.py:73-101, models/kan_alexnet.py:54-69).  Only the three
basis families on the accelerated path (plus the plain ``conv`` helper) are registered; the other
14 families of the reference are out of scope.
"""
from typing import Callable, List, Optional, Tuple, Union

import torch.nn as nn

from .conv_layers import ChebyKANConv2DLayer, FastKANConv2DLayer, KANConv2DLayer

_IntOrPair = Union[int, Tuple[int, int]]


def _calculate_same_padding(kernel_size: _IntOrPair, dilation: _IntOrPair) -> _IntOrPair:
    """'same' padding for stride 1: dilation*(k-1)//2 per axis (kan_conv.py:12-25)."""
    kh, kw = (kernel_size, kernel_size) if isinstance(kernel_size, int) else kernel_size
    dh, dw = (dilation, dilation) if isinstance(dilation, int) else dilation
    ph, pw = (dh * (kh - 1)) // 2, (dw * (kw - 1)) // 2
    return ph if (ph == pw and kh == kw) else (ph, pw)


def _no_l1(l1_decay: float):
    if l1_decay > 0:
        raise NotImplementedError("l1_decay > 0 wraps the layer in utils.regularization.L1 in the reference; that module is "
                                  "outside the accelerated path -- wrap the returned layer yourself")


def kan_conv(in_planes: int, out_planes: int, kernel_size: _IntOrPair, spline_order: int = 3, groups: int = 1,
             stride: _IntOrPair = 1, dilation: _IntOrPair = 1, padding: Optional[_IntOrPair] = None, grid_size: int = 5,
             base_activation: Optional[Callable[..., nn.Module]] = nn.GELU, grid_range: List = [-1, 1],
             l1_decay: float = 0.0, dropout: float = 0.0,
             norm_layer: Optional[Callable[..., nn.Module]] = nn.InstanceNorm2d, **norm_kwargs) -> KANConv2DLayer:
    if padding is None:
        padding = _calculate_same_padding(kernel_size, dilation)
    _no_l1(l1_decay)
    return KANConv2DLayer(input_dim=in_planes, output_dim=out_planes, kernel_size=kernel_size, spline_order=spline_order,
                          stride=stride, padding=padding, dilation=dilation, groups=groups, grid_size=grid_size,
                          base_activation=base_activation, grid_range=grid_range, dropout=dropout, norm_layer=norm_layer,
                          **norm_kwargs)


def chebykan_conv(in_planes: int, out_planes: int, kernel_size: _IntOrPair, degree: int = 3, groups: int = 1,
                  stride: _IntOrPair = 1, dilation: _IntOrPair = 1, padding: Optional[_IntOrPair] = None,
                  l1_decay: float = 0.0, dropout: float = 0.0,
                  norm_layer: Optional[Callable[..., nn.Module]] = nn.InstanceNorm2d, **norm_kwargs) -> ChebyKANConv2DLayer:
    if padding is None:
        padding = _calculate_same_padding(kernel_size, dilation)
    _no_l1(l1_decay)
    return ChebyKANConv2DLayer(input_dim=in_planes, output_dim=out_planes, kernel_size=kernel_size, degree=degree, stride=stride,
                               padding=padding, dilation=dilation, groups=groups, dropout=dropout, norm_layer=norm_layer,
                               **norm_kwargs)


def fastkan_conv(in_planes: int, out_planes: int, kernel_size: _IntOrPair, groups: int = 1, stride: _IntOrPair = 1,
                 dilation: _IntOrPair = 1, padding: Optional[_IntOrPair] = None, grid_size: int = 8,
                 base_activation: Callable[..., nn.Module] = nn.SiLU, grid_range: List = [-2, 2], l1_decay: float = 0.0,
                 dropout: float = 0.0, norm_layer: Optional[Callable[..., nn.Module]] = nn.InstanceNorm2d,
                 **norm_kwargs) -> FastKANConv2DLayer:
    if padding is None:
        padding = _calculate_same_padding(kernel_size, dilation)
    _no_l1(l1_decay)
    # the reference forwards l1_decay into **norm_kwargs, where the signature filter drops it (kan_conv.py:258-272)
    return FastKANConv2DLayer(input_dim=in_planes, output_dim=out_planes, kernel_size=kernel_size, stride=stride, padding=padding,
                              dilation=dilation, groups=groups, grid_size=grid_size, base_activation=base_activation,
                              grid_range=grid_range, dropout=dropout, l1_decay=l1_decay, norm_layer=norm_layer, **norm_kwargs)


def conv(in_planes: int, out_planes: int, kernel_size: _IntOrPair, groups: int = 1, stride: _IntOrPair = 1,
         dilation: _IntOrPair = 1, padding: Optional[_IntOrPair] = None,
         base_activation: Optional[Callable[..., nn.Module]] = nn.GELU,
         norm_layer: Optional[Callable[..., nn.Module]] = nn.BatchNorm2d, l1_decay: float = 0.0, dropout: float = 0.0,
         **kwargs) -> nn.Sequential:
    """Plain [Dropout] -> Conv2d -> [norm] -> [activation] block (kan_conv.py:71-116); not on the KAN path, torch ops only."""
    if padding is None:
        padding = _calculate_same_padding(kernel_size, dilation)
    _no_l1(l1_decay)
    mods = [nn.Dropout(p=dropout)] if dropout > 0 else []
    mods.append(nn.Conv2d(in_planes, out_planes, kernel_size, stride=stride, padding=padding, dilation=dilation, groups=groups,
                          bias=norm_layer is None))
    if norm_layer is not None:
        mods.append(norm_layer(out_planes))
    if base_activation is not None:
        mods.append(base_activation())
    return nn.Sequential(*mods)


CONV_KAN_FACTORY = {
    "KAN": kan_conv,
    "FastKAN": fastkan_conv,
    "ChebyKAN": chebykan_conv,
    "conv": conv,
}
