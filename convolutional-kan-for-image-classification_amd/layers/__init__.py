from .conv_layers import *          # noqa: F401,F403
from .kan_conv import CONV_KAN_FACTORY, kan_conv, fastkan_conv, chebykan_conv, conv   # noqa: F401
from .mlp_layers import KANLayer, KAN, mlp_kan, MLP_KAN_FACTORY   # noqa: F401
