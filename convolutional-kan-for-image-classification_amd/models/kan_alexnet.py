"""KAN-AlexNet caller (counterpart of the reference's models/kan_alexnet.py:10-313): plain FC head ('Linear' / 'AlexNet') or the
reference's 'KAN' head -- two Linear+ReLU stages and a B-spline MLP KAN as the last stage (`kan_fc3`, kan_alexnet.py:151-167,184-199)."""
import os
from functools import partial
from inspect import signature
from typing import Any, Callable, List, Optional

import torch
import torch.nn as nn

from ..layers.conv_layers import ChebyKANConvNDLayer, KANConvNDLayer
from ..layers.poly_layers import _RecurrenceKANConvNDLayer
from ..layers.kan_conv import CONV_KAN_FACTORY
from ..layers.mlp_layers import MLP_KAN_FACTORY


class AlexNetKAN(nn.Module):
    def __init__(self, num_classes: int = 1000, dropout: float = 0.5, input_channels: int = 3, arch: str = "default",
                 kan_conv: str = "KAN", classifier_type: str = "Linear", groups: int = 1, spline_order: int = 3, grid_size: int = 5,
                 base_activation: Optional[Callable[..., nn.Module]] = nn.SiLU, grid_range: List = [-1, 1],
                 degree: Optional[int] = 3, l1_decay: float = 0.0, affine: bool = True,
                 kan_norm_layer: Optional[Callable[..., nn.Module]] = nn.InstanceNorm2d, classifier_dropout: Optional[float] = None,
                 conv_dropout: float = 0.0, kan_classifier: Optional[str] = "KAN", classifier_spline_order: Optional[int] = None,
                 classifier_grid_size: Optional[int] = None, classifier_base_activation: Optional[Callable[..., nn.Module]] = None,
                 classifier_grid_range: Optional[List] = None, classifier_l1_decay: Optional[float] = None,
                 classifier_degree: Optional[int] = None, **kwargs: Any) -> None:
        super().__init__()
        if classifier_type in ("KAN", "AlexNetKAN") and (kan_classifier or "KAN") not in MLP_KAN_FACTORY:
            raise NotImplementedError(f"kan_classifier={kan_classifier!r}: only the B-spline MLP KAN head is built ({list(MLP_KAN_FACTORY)})")
        if kan_conv not in CONV_KAN_FACTORY:
            raise ValueError(f"kan_conv={kan_conv!r} is not on the accelerated path: {list(CONV_KAN_FACTORY)}")
        make = CONV_KAN_FACTORY[kan_conv]
        # kan_alexnet.py:54-69: these go in unfiltered, so e.g. `affine` reaches the layer's **norm_kwargs
        args = dict(spline_order=spline_order, grid_size=grid_size, base_activation=base_activation, grid_range=grid_range,
                    dropout=conv_dropout, l1_decay=l1_decay, groups=groups, norm_layer=kan_norm_layer, affine=affine, degree=degree)
        args.update({k: v for k, v in kwargs.items() if k in signature(make).parameters})
        block = partial(make, **args)
        self.arch = arch
        first = dict(kernel_size=11, stride=4, padding=2) if arch == "default" else dict(kernel_size=5, stride=1, padding=2)
        self.features = nn.Sequential(
            block(input_channels, 64, groups=groups, **first), nn.MaxPool2d(kernel_size=3, stride=2),
            block(64, 192, kernel_size=5, padding=2, groups=groups), nn.MaxPool2d(kernel_size=3, stride=2),
            block(192, 384, kernel_size=3, padding=1, groups=groups),
            block(384, 256, kernel_size=3, padding=1, groups=groups),
            block(256, 256, kernel_size=3, padding=1, groups=groups), nn.MaxPool2d(kernel_size=3, stride=2))
        self.avgpool = nn.AdaptiveAvgPool2d((6, 6))
        self.fuse_pool = os.environ.get("KAN_FUSE_POOL", "1") != "0"      # plain attribute: set False for the unfused sequence
        hid = 4096 if arch == "default" else 1024
        p = dropout if classifier_dropout is None else classifier_dropout
        if classifier_type == "KAN":
            # kan_alexnet.py:151-167: the head's KAN inherits the conv stage's spline settings unless overridden, takes the
            # classifier dropout, and is filtered by the factory's signature; only the LAST stage is a KAN (184-199)
            pick = lambda v, d: d if v is None else v
            offered = dict(dropout=p, spline_order=pick(classifier_spline_order, spline_order), grid_size=pick(classifier_grid_size, grid_size),
                           base_activation=pick(classifier_base_activation, base_activation), grid_range=pick(classifier_grid_range, grid_range),
                           l1_decay=pick(classifier_l1_decay, l1_decay), degree=pick(classifier_degree, degree), first_dropout=False)
            make_head = MLP_KAN_FACTORY[kan_classifier or "KAN"]
            last = ("kan_fc3", make_head(layers_hidden=[hid, num_classes], **{k: v for k, v in offered.items() if k in signature(make_head).parameters}))
        else:                                                     # 'Linear', 'AlexNet' and any other name: the plain head (168-183, 200-206)
            last = ("fc3", nn.Linear(hid, num_classes))
        self.classifier = nn.Sequential()
        for name, mod in (("head_dropout1", nn.Dropout(p=p)), ("fc1", nn.Linear(256 * 6 * 6, hid)), ("relu1", nn.ReLU(True)),
                          ("head_dropout2", nn.Dropout(p=p)), ("fc2", nn.Linear(hid, hid)), ("relu2", nn.ReLU(True)), last):
            self.classifier.add_module(name, mod)
        self._initialize_weights()
        head = classifier_type + (f"_{(kan_classifier or 'KAN').upper()}" if classifier_type in ("KAN", "AlexNetKAN") else "")
        self.name = f"AlexNet_{head}_{kan_conv.upper()}"

    def _initialize_weights(self) -> None:
        # kan_alexnet.py:236-250: every nn.Conv2d (including the KAN layers' weight holders) is re-initialised
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d, nn.GroupNorm)):
                if m.weight is not None:
                    nn.init.constant_(m.weight, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.constant_(m.bias, 0)

    def forward_features(self, x):
        """kan_alexnet.py:228 `self.features(x)`, with one fusion: a B-spline / ChebyKAN / recurrence-family layer directly followed by a plain
        MaxPool2d(k, s) (here (3, 2)) runs the pooling inside its InstanceNorm(+PReLU) kernels -- the un-pooled activation and its gradient never
        go through HBM and torch's two pooling kernels disappear.  Same values as the two modules in sequence (tests/test_gpu_models.py)."""
        mods = list(self.features)
        i = 0
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            ks = _plain_pool(nxt) if (self.fuse_pool and isinstance(nxt, nn.MaxPool2d)) else None
            if ks is not None and isinstance(m, (KANConvNDLayer, ChebyKANConvNDLayer, _RecurrenceKANConvNDLayer)) and getattr(m, "ndim", 2) == 2:
                x = m(x, pool=ks)
                i += 2
            else:
                x = m(x)
                i += 1
        return x

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.avgpool(self.forward_features(x))
        return self.classifier(torch.flatten(x, 1))


def _plain_pool(m: nn.MaxPool2d):
    """(kernel, stride) of a square, unpadded, undilated floor-mode MaxPool2d the norm kernels can fuse, else None."""
    sq = lambda v: (v if isinstance(v, int) else v[0] if (len(v) == 2 and v[0] == v[1]) else None)
    k, st = sq(m.kernel_size), sq(m.stride if m.stride is not None else m.kernel_size)
    zero = m.padding == 0 or m.padding == (0, 0)
    one = m.dilation == 1 or m.dilation == (1, 1)
    if k is None or st is None or not zero or not one or m.ceil_mode or m.return_indices or not (2 <= k <= 15 and 1 <= st <= k):
        return None
    return (k, st)


def alexnet_kan(num_classes: int = 1000, input_channels: int = 3, dropout: float = 0.5, arch: str = "default",
                conv_type: str = "kanconv", kan_conv: Optional[str] = "KAN", classifier_type: str = "Linear", **kwargs: Any) -> AlexNetKAN:
    if conv_type != "kanconv":
        raise NotImplementedError("only conv_type='kanconv' is on the accelerated path")
    return AlexNetKAN(num_classes=num_classes, dropout=dropout, input_channels=input_channels, arch=arch,
                      kan_conv=kan_conv or "KAN", classifier_type=classifier_type, **kwargs)
