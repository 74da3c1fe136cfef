from .kan_vgg import VGGKAN, vggkan, cfgs          # noqa: F401
from .kan_alexnet import AlexNetKAN, alexnet_kan   # noqa: F401
