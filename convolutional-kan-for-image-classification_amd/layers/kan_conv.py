"""Factory functions with the reference's signatures (layers/kan_conv.py:27-69, 197-276, 726-745).

``CONV_KAN_FACTORY[name](in_planes, out_planes, kernel_size=..., ...)`` is the drop-in boundary the
reference's models use (models/kan_vgg.py:73-101, models/kan_alexnet.py:54-69).  Registered: the three
basis families of the hot path, the eight three-term-recurrence polynomial families of SURVEY.md
section 8(f) rank 3 (Bessel, Fibonacci, Gegenbauer, Hermite, Jacobi, Laguerre, Lucas, Taylor), FourierKAN and the
plain ``conv`` helper, plus LegendreKAN, BersnsteinKAN, ReLU-KAN and GRAM-KAN (trainable parameters inside the basis:
device-side phase / coefficient tables) and Wav-KAN (per-(output, input) wavelets on the direct kernels of csrc/wavkan.inc) --
all 18 of the reference's keys.
"""
from typing import Callable, List, Optional, Tuple, Union

import torch.nn as nn

from .conv_layers import ChebyKANConv2DLayer, FastKANConv2DLayer, KANConv2DLayer
from .relu_layers import ReLUKANConv2DLayer
from ..utils.regularization import L1
from .gram_layers import GRAMKANConv2DLayer
from .poly_layers import (BersnsteinKANConv2DLayer, BesselKANConv2DLayer, FibonacciKANConv2DLayer, FourierKANConv2DLayer, LegendreKANConv2DLayer, GegenbauerKANConv2DLayer, HermiteKANConv2DLayer,
                          JacobiKANConv2DLayer, LaguerreKANConv2DLayer, LucasKANConv2DLayer, TaylorKANConv2DLayer)

from .wav_layers import WavKANConv2DLayer

_IntOrPair = Union[int, Tuple[int, int]]


def _calculate_same_padding(kernel_size: _IntOrPair, dilation: _IntOrPair) -> _IntOrPair:
    """'same' padding for stride 1: dilation*(k-1)//2 per axis (kan_conv.py:12-25)."""
    kh, kw = (kernel_size, kernel_size) if isinstance(kernel_size, int) else kernel_size
    dh, dw = (dilation, dilation) if isinstance(dilation, int) else dilation
    ph, pw = (dh * (kh - 1)) // 2, (dw * (kw - 1)) // 2
    return ph if (ph == pw and kh == kw) else (ph, pw)


def _l1(l1_decay: float, layer: nn.Module) -> nn.Module:
    """layers/kan_conv.py:66-68 (and every sibling factory): `if l1_decay > 0: conv = L1(conv, l1_decay)`."""
    return L1(layer, l1_decay) if l1_decay > 0 else layer


def kan_conv(in_planes: int, out_planes: int, kernel_size: _IntOrPair, spline_order: int = 3, groups: int = 1,
             stride: _IntOrPair = 1, dilation: _IntOrPair = 1, padding: Optional[_IntOrPair] = None, grid_size: int = 5,
             base_activation: Optional[Callable[..., nn.Module]] = nn.GELU, grid_range: List = [-1, 1],
             l1_decay: float = 0.0, dropout: float = 0.0,
             norm_layer: Optional[Callable[..., nn.Module]] = nn.InstanceNorm2d, **norm_kwargs) -> KANConv2DLayer:
    if padding is None:
        padding = _calculate_same_padding(kernel_size, dilation)
    return _l1(l1_decay, KANConv2DLayer(input_dim=in_planes, output_dim=out_planes, kernel_size=kernel_size, spline_order=spline_order,
                          stride=stride, padding=padding, dilation=dilation, groups=groups, grid_size=grid_size,
                          base_activation=base_activation, grid_range=grid_range, dropout=dropout, norm_layer=norm_layer,
                          **norm_kwargs))


def chebykan_conv(in_planes: int, out_planes: int, kernel_size: _IntOrPair, degree: int = 3, groups: int = 1,
                  stride: _IntOrPair = 1, dilation: _IntOrPair = 1, padding: Optional[_IntOrPair] = None,
                  l1_decay: float = 0.0, dropout: float = 0.0,
                  norm_layer: Optional[Callable[..., nn.Module]] = nn.InstanceNorm2d, **norm_kwargs) -> ChebyKANConv2DLayer:
    if padding is None:
        padding = _calculate_same_padding(kernel_size, dilation)
    return _l1(l1_decay, ChebyKANConv2DLayer(input_dim=in_planes, output_dim=out_planes, kernel_size=kernel_size, degree=degree, stride=stride,
                               padding=padding, dilation=dilation, groups=groups, dropout=dropout, norm_layer=norm_layer,
                               **norm_kwargs))


def fastkan_conv(in_planes: int, out_planes: int, kernel_size: _IntOrPair, groups: int = 1, stride: _IntOrPair = 1,
                 dilation: _IntOrPair = 1, padding: Optional[_IntOrPair] = None, grid_size: int = 8,
                 base_activation: Callable[..., nn.Module] = nn.SiLU, grid_range: List = [-2, 2], l1_decay: float = 0.0,
                 dropout: float = 0.0, norm_layer: Optional[Callable[..., nn.Module]] = nn.InstanceNorm2d,
                 **norm_kwargs) -> FastKANConv2DLayer:
    if padding is None:
        padding = _calculate_same_padding(kernel_size, dilation)
    # the reference also forwards l1_decay into **norm_kwargs, where the signature filter drops it (kan_conv.py:258-272)
    return _l1(l1_decay, FastKANConv2DLayer(input_dim=in_planes, output_dim=out_planes, kernel_size=kernel_size, stride=stride,
                                            padding=padding, dilation=dilation, groups=groups, grid_size=grid_size,
                                            base_activation=base_activation, grid_range=grid_range, dropout=dropout, l1_decay=l1_decay,
                                            norm_layer=norm_layer, **norm_kwargs))


def conv(in_planes: int, out_planes: int, kernel_size: _IntOrPair, groups: int = 1, stride: _IntOrPair = 1,
         dilation: _IntOrPair = 1, padding: Optional[_IntOrPair] = None,
         base_activation: Optional[Callable[..., nn.Module]] = nn.GELU,
         norm_layer: Optional[Callable[..., nn.Module]] = nn.BatchNorm2d, l1_decay: float = 0.0, dropout: float = 0.0,
         **kwargs) -> nn.Sequential:
    """Plain [Dropout] -> Conv2d -> [norm] -> [activation] block (kan_conv.py:71-116); not on the KAN path, torch ops only."""
    if padding is None:
        padding = _calculate_same_padding(kernel_size, dilation)
    mods = [nn.Dropout(p=dropout)] if dropout > 0 else []
    mods.append(_l1(l1_decay, nn.Conv2d(in_planes, out_planes, kernel_size, stride=stride, padding=padding, dilation=dilation,
                                        groups=groups, bias=norm_layer is None)))          # kan_conv.py:105-108: L1 around the Conv2d only
    if norm_layer is not None:
        mods.append(norm_layer(out_planes))
    if base_activation is not None:
        mods.append(base_activation())
    return nn.Sequential(*mods)


# ---- three-term-recurrence polynomial families.  As in the reference (kan_conv.py:354-724) `dilation` only enters the
# 'same' padding and is NOT forwarded to the layer, and `l1_decay` travels in **norm_kwargs where the signature filter
# drops it (Taylor: not forwarded at all).
def legendrekan_conv(in_planes: int, out_planes: int, kernel_size: _IntOrPair, degree: int = 3, groups: int = 1, stride: _IntOrPair = 1,
                     dilation: _IntOrPair = 1, padding: Optional[_IntOrPair] = None, dropout: float = 0.0,
                     norm_layer: Optional[Callable[..., nn.Module]] = nn.InstanceNorm2d, l1_decay: float = 0.0,
                     **norm_kwargs) -> LegendreKANConv2DLayer:
    """kan_conv.py:120-155 (this one does forward `dilation`)."""
    if padding is None:
        padding = _calculate_same_padding(kernel_size, dilation)
    return _l1(l1_decay, LegendreKANConv2DLayer(input_dim=in_planes, output_dim=out_planes, kernel_size=kernel_size, degree=degree, stride=stride,
                                  padding=padding, dilation=dilation, groups=groups, dropout=dropout, norm_layer=norm_layer, **norm_kwargs))


def bersnsteinkan_conv(in_planes: int, out_planes: int, kernel_size: _IntOrPair, groups: int = 1, stride: _IntOrPair = 1,
                       dilation: _IntOrPair = 1, padding: Optional[_IntOrPair] = None, l1_decay: float = 0.0, dropout: float = 0.0,
                       degree: int = 3, norm_layer: Optional[Callable[..., nn.Module]] = nn.InstanceNorm2d,
                       **norm_kwargs) -> BersnsteinKANConv2DLayer:
    """kan_conv.py:319-351 (forwards `dilation`; `l1_decay` rides in **norm_kwargs and is filtered out)."""
    if padding is None:
        padding = _calculate_same_padding(kernel_size, dilation)
    return _l1(l1_decay, BersnsteinKANConv2DLayer(input_dim=in_planes, output_dim=out_planes, kernel_size=kernel_size, degree=degree, groups=groups,
                                    stride=stride, padding=padding, dilation=dilation, dropout=dropout, l1_decay=l1_decay,
                                    norm_layer=norm_layer, **norm_kwargs))


def besselkan_conv(in_planes: int, out_planes: int, kernel_size: _IntOrPair, groups: int = 1, stride: _IntOrPair = 1,
                   dilation: _IntOrPair = 1, padding: Optional[_IntOrPair] = None, l1_decay: float = 0.0, dropout: float = 0.0,
                   degree: int = 3, base_activation: Optional[Callable[..., nn.Module]] = nn.GELU,
                   norm_layer: Optional[Callable[..., nn.Module]] = nn.InstanceNorm2d, **norm_kwargs) -> BesselKANConv2DLayer:
    if padding is None:
        padding = _calculate_same_padding(kernel_size, dilation)
    return _l1(l1_decay, BesselKANConv2DLayer(input_dim=in_planes, output_dim=out_planes, kernel_size=kernel_size, degree=degree, groups=groups,
                                padding=padding, stride=stride, l1_decay=l1_decay, dropout=dropout, base_activation=base_activation,
                                norm_layer=norm_layer, **norm_kwargs))


def fibonaccikan_conv(in_planes: int, out_planes: int, kernel_size: _IntOrPair, groups: int = 1, stride: _IntOrPair = 1,
                      dilation: _IntOrPair = 1, padding: Optional[_IntOrPair] = None, l1_decay: float = 0.0, dropout: float = 0.0,
                      degree: int = 3, base_activation: Optional[Callable[..., nn.Module]] = nn.GELU,
                      norm_layer: Optional[Callable[..., nn.Module]] = nn.InstanceNorm2d, **norm_kwargs) -> FibonacciKANConv2DLayer:
    if padding is None:
        padding = _calculate_same_padding(kernel_size, dilation)
    return _l1(l1_decay, FibonacciKANConv2DLayer(input_dim=in_planes, output_dim=out_planes, kernel_size=kernel_size, degree=degree, groups=groups,
                                   padding=padding, stride=stride, l1_decay=l1_decay, dropout=dropout, base_activation=base_activation,
                                   norm_layer=norm_layer, **norm_kwargs))


def fourierkan_conv(in_planes: int, out_planes: int, kernel_size: _IntOrPair, groups: int = 1, stride: _IntOrPair = 1,
                    dilation: _IntOrPair = 1, padding: Optional[_IntOrPair] = None, l1_decay: float = 0.0, dropout: float = 0.0,
                    grid_size: int = 3, base_activation: Optional[Callable[..., nn.Module]] = nn.GELU,
                    norm_layer: Optional[Callable[..., nn.Module]] = nn.InstanceNorm2d, **norm_kwargs) -> FourierKANConv2DLayer:
    if padding is None:
        padding = _calculate_same_padding(kernel_size, dilation)
    return _l1(l1_decay, FourierKANConv2DLayer(input_dim=in_planes, output_dim=out_planes, kernel_size=kernel_size, grid_size=grid_size, groups=groups,
                                 padding=padding, stride=stride, l1_decay=l1_decay, dropout=dropout, base_activation=base_activation,
                                 norm_layer=norm_layer, **norm_kwargs))


def gegenbauerkan_conv(in_planes: int, out_planes: int, kernel_size: _IntOrPair, groups: int = 1, stride: _IntOrPair = 1,
                       dilation: _IntOrPair = 1, padding: Optional[_IntOrPair] = None, l1_decay: float = 0.0, dropout: float = 0.0,
                       degree: int = 3, alpha_param: float = 0.0, base_activation: Optional[Callable[..., nn.Module]] = nn.GELU,
                       norm_layer: Optional[Callable[..., nn.Module]] = nn.InstanceNorm2d, **norm_kwargs) -> GegenbauerKANConv2DLayer:
    if padding is None:
        padding = _calculate_same_padding(kernel_size, dilation)
    return _l1(l1_decay, GegenbauerKANConv2DLayer(input_dim=in_planes, output_dim=out_planes, kernel_size=kernel_size, degree=degree, groups=groups,
                                    padding=padding, stride=stride, l1_decay=l1_decay, dropout=dropout, alpha_param=alpha_param,
                                    base_activation=base_activation, norm_layer=norm_layer, **norm_kwargs))


def hermitekan_conv(in_planes: int, out_planes: int, kernel_size: _IntOrPair, groups: int = 1, stride: _IntOrPair = 1,
                    dilation: _IntOrPair = 1, padding: Optional[_IntOrPair] = None, l1_decay: float = 0.0, dropout: float = 0.0,
                    degree: int = 3, base_activation: Optional[Callable[..., nn.Module]] = nn.GELU,
                    norm_layer: Optional[Callable[..., nn.Module]] = nn.InstanceNorm2d, **norm_kwargs) -> HermiteKANConv2DLayer:
    if padding is None:
        padding = _calculate_same_padding(kernel_size, dilation)
    return _l1(l1_decay, HermiteKANConv2DLayer(input_dim=in_planes, output_dim=out_planes, kernel_size=kernel_size, degree=degree, groups=groups,
                                 padding=padding, stride=stride, l1_decay=l1_decay, dropout=dropout, base_activation=base_activation,
                                 norm_layer=norm_layer, **norm_kwargs))


def jacobikan_conv(in_planes: int, out_planes: int, kernel_size: _IntOrPair, groups: int = 1, stride: _IntOrPair = 1,
                   dilation: _IntOrPair = 1, padding: Optional[_IntOrPair] = None, l1_decay: float = 0.0, dropout: float = 0.0,
                   degree: int = 3, a: float = 1.0, b: float = 1.0, base_activation: Optional[Callable[..., nn.Module]] = nn.GELU,
                   norm_layer: Optional[Callable[..., nn.Module]] = nn.InstanceNorm2d, **norm_kwargs) -> JacobiKANConv2DLayer:
    if padding is None:
        padding = _calculate_same_padding(kernel_size, dilation)
    return _l1(l1_decay, JacobiKANConv2DLayer(input_dim=in_planes, output_dim=out_planes, kernel_size=kernel_size, degree=degree, a=a, b=b,
                                groups=groups, padding=padding, stride=stride, l1_decay=l1_decay, dropout=dropout,
                                base_activation=base_activation, norm_layer=norm_layer, **norm_kwargs))


def relukan_conv(in_planes: int, out_planes: int, kernel_size: _IntOrPair, groups: int = 1, stride: _IntOrPair = 1,
                 dilation: _IntOrPair = 1, padding: Optional[_IntOrPair] = None, l1_decay: float = 0.0, dropout: float = 0.0,
                 g: int = 5, k: int = 3, train_ab: bool = True, base_activation: Optional[Callable[..., nn.Module]] = nn.GELU,
                 norm_layer: Optional[Callable[..., nn.Module]] = nn.InstanceNorm2d, **norm_kwargs) -> ReLUKANConv2DLayer:
    """layers/kan_conv.py:652-690.  As there, `dilation` only enters the 'same' padding: the layer itself is built with the
    default dilation 1."""
    if padding is None:
        padding = _calculate_same_padding(kernel_size, dilation)
    return _l1(l1_decay, ReLUKANConv2DLayer(input_dim=in_planes, output_dim=out_planes, kernel_size=kernel_size, g=g, k=k, train_ab=train_ab,
                              groups=groups, padding=padding, stride=stride, l1_decay=l1_decay, dropout=dropout,
                              base_activation=base_activation, norm_layer=norm_layer, **norm_kwargs))


def gramkan_conv(in_planes: int, out_planes: int, kernel_size: _IntOrPair, degree: int = 3, groups: int = 1, stride: _IntOrPair = 1,
                 dilation: _IntOrPair = 1, padding: Optional[_IntOrPair] = None, dropout: float = 0.0,
                 norm_layer: Optional[Callable[..., nn.Module]] = nn.InstanceNorm2d, l1_decay: float = 0.0,
                 **norm_kwargs) -> GRAMKANConv2DLayer:
    """layers/kan_conv.py:158-194 (no base_activation argument: the 2-D GRAM layer always runs SiLU)."""
    if padding is None:
        padding = _calculate_same_padding(kernel_size, dilation)
    return _l1(l1_decay, GRAMKANConv2DLayer(input_dim=in_planes, output_dim=out_planes, kernel_size=kernel_size, degree=degree, stride=stride,
                              padding=padding, dilation=dilation, groups=groups, dropout=dropout, norm_layer=norm_layer, **norm_kwargs))


def laguerrekan_conv(in_planes: int, out_planes: int, kernel_size: _IntOrPair, groups: int = 1, stride: _IntOrPair = 1,
                     dilation: _IntOrPair = 1, padding: Optional[_IntOrPair] = None, l1_decay: float = 0.0, dropout: float = 0.0,
                     degree: int = 3, alpha: float = 1.0, base_activation: Optional[Callable[..., nn.Module]] = nn.GELU,
                     norm_layer: Optional[Callable[..., nn.Module]] = nn.InstanceNorm2d, **norm_kwargs) -> LaguerreKANConv2DLayer:
    if padding is None:
        padding = _calculate_same_padding(kernel_size, dilation)
    return _l1(l1_decay, LaguerreKANConv2DLayer(input_dim=in_planes, output_dim=out_planes, kernel_size=kernel_size, degree=degree, alpha=alpha,
                                  groups=groups, padding=padding, stride=stride, l1_decay=l1_decay, dropout=dropout,
                                  base_activation=base_activation, norm_layer=norm_layer, **norm_kwargs))


def lucaskan_conv(in_planes: int, out_planes: int, kernel_size: _IntOrPair, groups: int = 1, stride: _IntOrPair = 1,
                  dilation: _IntOrPair = 1, padding: Optional[_IntOrPair] = None, l1_decay: float = 0.0, dropout: float = 0.0,
                  degree: int = 3, base_activation: Optional[Callable[..., nn.Module]] = nn.GELU,
                  norm_layer: Optional[Callable[..., nn.Module]] = nn.InstanceNorm2d, **norm_kwargs) -> LucasKANConv2DLayer:
    if padding is None:
        padding = _calculate_same_padding(kernel_size, dilation)
    return _l1(l1_decay, LucasKANConv2DLayer(input_dim=in_planes, output_dim=out_planes, kernel_size=kernel_size, degree=degree, groups=groups,
                               padding=padding, stride=stride, l1_decay=l1_decay, dropout=dropout, base_activation=base_activation,
                               norm_layer=norm_layer, **norm_kwargs))


def taylorkan_conv(in_planes: int, out_planes: int, kernel_size: _IntOrPair, groups: int = 1, stride: _IntOrPair = 1,
                   dilation: _IntOrPair = 1, padding: Optional[_IntOrPair] = None, l1_decay: float = 0.0, dropout: float = 0.0,
                   degree: int = 3, base_activation: Optional[Callable[..., nn.Module]] = nn.GELU,
                   norm_layer: Optional[Callable[..., nn.Module]] = nn.InstanceNorm2d, **norm_kwargs) -> TaylorKANConv2DLayer:
    if padding is None:
        padding = _calculate_same_padding(kernel_size, dilation)
    return _l1(l1_decay, TaylorKANConv2DLayer(input_dim=in_planes, output_dim=out_planes, kernel_size=kernel_size, degree=degree, groups=groups,
                                padding=padding, stride=stride, dropout=dropout, base_activation=base_activation, norm_layer=norm_layer,
                                **norm_kwargs))



def wavkan_conv(in_planes: int, out_planes: int, kernel_size: _IntOrPair, groups: int = 1, stride: _IntOrPair = 1,
                dilation: _IntOrPair = 1, padding: Optional[_IntOrPair] = None, l1_decay: float = 0.0, dropout: float = 0.0,
                wavelet_type: str = 'mexican_hat', wav_version: str = 'fast',
                norm_layer: Optional[Callable[..., nn.Module]] = nn.InstanceNorm2d, **norm_kwargs) -> WavKANConv2DLayer:
    """kan_conv.py:278-318 (`l1_decay` travels into the layer's **norm_kwargs, where the norm-signature filter drops it)."""
    if padding is None:
        padding = _calculate_same_padding(kernel_size, dilation)
    return _l1(l1_decay, WavKANConv2DLayer(input_dim=in_planes, output_dim=out_planes, kernel_size=kernel_size, stride=stride, padding=padding,
                                           dilation=dilation, groups=groups, wavelet_type=wavelet_type, wav_version=wav_version, dropout=dropout,
                                           l1_decay=l1_decay, norm_layer=norm_layer, **norm_kwargs))


CONV_KAN_FACTORY = {
    "KAN": kan_conv,
    "FastKAN": fastkan_conv,
    "GRAMKAN": gramkan_conv,
    "ChebyKAN": chebykan_conv,
    "LegendreKAN": legendrekan_conv,
    "BersnsteinKAN": bersnsteinkan_conv,
    "BesselKAN": besselkan_conv,
    "FibonacciKAN": fibonaccikan_conv,
    "FourierKAN": fourierkan_conv,
    "GegenbauerKAN": gegenbauerkan_conv,
    "HermiteKAN": hermitekan_conv,
    "JacobiKAN": jacobikan_conv,
    "LaguerreKAN": laguerrekan_conv,
    "LucasKAN": lucaskan_conv,
    "ReLUKAN": relukan_conv,
    "TaylorKAN": taylorkan_conv,
    "WavKAN": wavkan_conv,
    "conv": conv,
}
