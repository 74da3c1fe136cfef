"""convkan_amd: MI355X-native conv-KAN layers (B-spline, FastKAN/RBF, Chebyshev) behind the
constructor / factory API of GadGadGad/Convolutional-KAN-for-Image-Classification.

Import as ``convkan_amd`` (the repo-root shim maps that name onto this directory, whose
on-disk name ``convolutional-kan-for-image-classification_amd`` is not a Python identifier).
"""
from . import _lib, ops                                                   # noqa: F401
from .layers import (CONV_KAN_FACTORY, KANConv2DLayer, KANConvNDLayer, FastKANConv2DLayer, FastKANConvNDLayer,   # noqa: F401
                     ChebyKANConv2DLayer, ChebyKANConvNDLayer, RadialBasisFunction, kan_conv, fastkan_conv, chebykan_conv, conv,
                     KANLayer, KAN, mlp_kan, MLP_KAN_FACTORY)
from .build import build_library                                          # noqa: F401

__version__ = "0.1.0"
from .layers.poly_layers import (LegendreKANConv2DLayer, BersnsteinKANConv2DLayer, FourierKANConv2DLayer, BesselKANConv2DLayer, FibonacciKANConv2DLayer, GegenbauerKANConv2DLayer, HermiteKANConv2DLayer,   # noqa: F401,E402
                                 JacobiKANConv2DLayer, LaguerreKANConv2DLayer, LucasKANConv2DLayer, TaylorKANConv2DLayer)
from .layers.conv_layers import KANConv1DLayer, FastKANConv1DLayer, ChebyKANConv1DLayer   # noqa: F401,E402
from .layers.relu_layers import ReLUConvNDLayer, ReLUKANConv2DLayer, ReLUKANConv1DLayer   # noqa: F401,E402
from .layers.kan_conv import relukan_conv   # noqa: F401,E402
from .optim import FusedAdamW   # noqa: F401,E402
from .train import GraphedStep, train_step, train_model_generic   # noqa: F401,E402
from .layers.gram_layers import GRAMKANConvNDLayer, GRAMKANConv2DLayer   # noqa: F401,E402
from .layers.kan_conv import gramkan_conv   # noqa: F401,E402
from .layers.conv_layers import KANConv3DLayer, FastKANConv3DLayer, ChebyKANConv3DLayer   # noqa: F401,E402
from .layers.poly_layers import (BesselKANConv1DLayer, FibonacciKANConv1DLayer, GegenbauerKANConv1DLayer, HermiteKANConv1DLayer,   # noqa: F401,E402
                                 LaguerreKANConv1DLayer, LucasKANConv1DLayer, TaylorKANConv1DLayer, FourierKANConv1DLayer)
from .layers.wav_layers import (WaveletConvND, WaveletConvNDFast, WaveletConvNDFastPlusOne, WavKANConvNDLayer, WavKANConv2DLayer,   # noqa: F401,E402
                                WavKANConv1DLayer, WavKANConv3DLayer)
from .layers.kan_conv import wavkan_conv   # noqa: F401,E402
from .layers.poly_layers import (BesselKANConv3DLayer, FibonacciKANConv3DLayer, GegenbauerKANConv3DLayer, HermiteKANConv3DLayer,   # noqa: F401,E402
                                 LaguerreKANConv3DLayer, LucasKANConv3DLayer, TaylorKANConv3DLayer, FourierKANConv3DLayer)
