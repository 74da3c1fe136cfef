from .conv_layers import *          # noqa: F401,F403
from .poly_layers import (LegendreKANConvNDLayer, LegendreKANConv2DLayer, BersnsteinKANConvNDLayer, BersnsteinKANConv2DLayer,
                          FourierKANConvNDLayer, FourierKANConv2DLayer, BesselKANConvNDLayer, BesselKANConv2DLayer, FibonacciKANConvNDLayer, FibonacciKANConv2DLayer,   # noqa: F401
                          GegenbauerKANConvNDLayer, GegenbauerKANConv2DLayer, HermiteKANConvNDLayer, HermiteKANConv2DLayer,
                          JacobiKANConvNDLayer, JacobiKANConv2DLayer, LaguerreKANConvNDLayer, LaguerreKANConv2DLayer,
                          LucasKANConvNDLayer, LucasKANConv2DLayer, TaylorKANConvNDLayer, TaylorKANConv2DLayer)
from .relu_layers import ReLUConvNDLayer, ReLUKANConv2DLayer, ReLUKANConv1DLayer   # noqa: F401
from .gram_layers import GRAMKANConvNDLayer, GRAMKANConv2DLayer   # noqa: F401
from .wav_layers import (WaveletConvND, WaveletConvNDFast, WaveletConvNDFastPlusOne, WavKANConvNDLayer, WavKANConv2DLayer, WavKANConv1DLayer, WavKANConv3DLayer)   # noqa: F401
from .kan_conv import (CONV_KAN_FACTORY, relukan_conv, gramkan_conv, kan_conv, fastkan_conv, chebykan_conv, conv, legendrekan_conv, bersnsteinkan_conv, besselkan_conv, fibonaccikan_conv, fourierkan_conv,   # noqa: F401
                       gegenbauerkan_conv, hermitekan_conv, jacobikan_conv, laguerrekan_conv, lucaskan_conv, taylorkan_conv, wavkan_conv)
from .mlp_layers import KANLayer, KAN, mlp_kan, MLP_KAN_FACTORY   # noqa: F401
