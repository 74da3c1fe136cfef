"""KAN-VGG caller built on CONV_KAN_FACTORY (counterpart of the reference's models/kan_vgg.py).

Same module tree and parameter names as the reference (``features.N.<layer>``, ``classifier.1``),
so a reference ``state_dict`` loads unchanged.  Adds the ``VGG11`` cfg that BASELINE.json names
(torchvision cfg "A" without its trailing "M" -- the reference's own convention for VGG16/19,
kan_vgg.py:20-26).  Only plain heads ('Linear', 'VGG') are offered: KAN MLP heads are out of the
accelerated path's first "next" row and are built on the same conv stage.
"""
import os
from functools import partial
from inspect import signature
from math import prod
from typing import Any, Callable, Dict, List, Optional, Tuple, Union

import torch
import torch.nn as nn

from ..layers.conv_layers import KANConvNDLayer
from ..layers.poly_layers import _RecurrenceKANConvNDLayer
from ..layers.kan_conv import CONV_KAN_FACTORY
from ..layers.mlp_layers import MLP_KAN_FACTORY

cfgs: Dict[str, List[Union[str, int]]] = {
    "VGG11": [64, "M", 128, "M", 256, 256, "M", 512, 512, "M", 512, 512],
    "VGG16_small": [16, 16, "M", 32, 32, "M", 64, 64, 64, "M", 128, 128, 128, "M", 128, 128, 128],
    "VGG16_kansmall": [8, 8, "M", 16, 16, "M", 32, 32, 32, "M", 64, 64, 64, "M", 64, 64, 64],
    "VGG19_small": [16, 16, "M", 32, 32, "M", 64, 64, 64, 64, "M", 128, 128, 128, 128, "M", 128, 128, 128, 128],
    "VGG16": [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512],
    "VGG19": [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512],
}


def _head(kind: str, feat: int, num_classes: int, p: float, kan_head=None) -> nn.Module:
    if kind == "Linear":                                        # kan_vgg.py:139-143
        return nn.Sequential(nn.Dropout(p=p), nn.Linear(feat, num_classes))
    if kind == "VGG":                                           # kan_vgg.py:164-173
        return nn.Sequential(nn.Linear(feat, 1024), nn.ReLU(True), nn.Dropout(p=p), nn.Linear(1024, 1024), nn.ReLU(True),
                             nn.Dropout(p=p), nn.Linear(1024, num_classes))
    if kind == "KAN":                                           # kan_vgg.py:134-138
        return nn.Sequential(nn.Dropout(p=p), kan_head([feat, num_classes]))
    if kind == "HiddenKAN":                                     # kan_vgg.py:144-149
        return nn.Sequential(kan_head([feat, 1024]), nn.Dropout(p=p), nn.Linear(1024, num_classes))
    if kind == "VGGKAN":                                        # kan_vgg.py:150-163
        return nn.Sequential(nn.Linear(feat, 1024), nn.ReLU(True), nn.Dropout(p=p), nn.Linear(1024, 1024), nn.ReLU(True),
                             nn.Dropout(p=p), kan_head([1024, num_classes]))
    raise NotImplementedError(f"classifier_type={kind!r} is not offered (Linear, VGG, KAN, HiddenKAN, VGGKAN)")


class VGGKAN(nn.Module):
    def __init__(self, input_channels: int, num_classes: int, kan_conv: str = "KAN", arch: str = "VGG16",
                 classifier_type: str = "Linear", groups: int = 1, spline_order: int = 3, grid_size: int = 5,
                 base_activation: Optional[Callable[..., nn.Module]] = nn.SiLU, grid_range: List = [-1, 1],
                 l1_decay: float = 0.0, dropout_linear: float = 0.5, expected_feature_shape: Tuple[int, int] = (1, 1),
                 width_scale: int = 1, affine: bool = False, kan_norm_layer: Optional[Callable[..., nn.Module]] = nn.InstanceNorm2d,
                 std_conv_kernel_size: int = 3, std_conv_padding: int = 1, degree: int = 3, conv_dropout: float = 0.0,
                 kan_classifier: Optional[str] = "KAN", **kwargs: Any):
        super().__init__()
        if arch not in cfgs:
            raise ValueError(f"Unknown arch: {arch}. Available types: {list(cfgs.keys())}")
        if kan_conv not in CONV_KAN_FACTORY:
            raise ValueError(f"kan_conv={kan_conv!r} is not on the accelerated path: {list(CONV_KAN_FACTORY)}")
        make = CONV_KAN_FACTORY[kan_conv]
        accepted = signature(make).parameters                  # kan_vgg.py:91-94: pass a superset, keep what the factory takes
        offered = dict(spline_order=spline_order, grid_size=grid_size, base_activation=base_activation, grid_range=grid_range,
                       l1_decay=l1_decay, dropout=conv_dropout, degree=degree, affine=affine, norm_layer=kan_norm_layer,
                       padding=std_conv_padding, groups=groups, kernel_size=std_conv_kernel_size)
        kw = {k: v for k, v in offered.items() if k in accepted}
        kw.update({k: v for k, v in kwargs.items() if k in accepted})
        first_kw = dict(kw, dropout=0.0)                       # first conv never drops (kan_vgg.py:99-101)

        feats: List[nn.Module] = []
        cin = input_channels
        for i, v in enumerate(cfgs[arch]):
            if v == "M":
                feats.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                cout = int(v) * width_scale
                feats.append(partial(make, **(first_kw if i == 0 else kw))(cin, cout))
                cin = cout
        self.features = nn.ModuleList(feats)
        self.avgpool = nn.AdaptiveAvgPool2d(expected_feature_shape)
        self.fuse_pool = os.environ.get("KAN_FUSE_POOL", "1") != "0"      # plain attribute: set False for the unfused sequence
        kan_head = None
        if classifier_type in ("KAN", "HiddenKAN", "VGGKAN"):
            name = kan_classifier or "KAN"
            if name not in MLP_KAN_FACTORY:
                raise NotImplementedError(f"kan_classifier={name!r}: only the B-spline MLP KAN head is built ({list(MLP_KAN_FACTORY)})")
            # kan_vgg.py:258-283: the head inherits spline_order / grid_size / grid_range, SiLU base, no dropout, filtered by signature
            offered_h = dict(spline_order=spline_order, grid_size=grid_size, base_activation=nn.SiLU, grid_range=grid_range,
                             l1_decay=l1_decay, degree=degree, dropout=0.0, first_dropout=False, bias=False)
            hk = {k: v for k, v in offered_h.items() if k in signature(MLP_KAN_FACTORY[name]).parameters}
            kan_head = partial(MLP_KAN_FACTORY[name], **hk)
        self.classifier = _head(classifier_type, cin * prod(expected_feature_shape), num_classes, dropout_linear, kan_head)
        self.expected_feature_shape = expected_feature_shape
        head = classifier_type + (f"_{(kan_classifier or 'KAN').upper()}" if kan_head is not None else "")
        self.name = f"VGGKAN_{head}_{kan_conv.upper()}_{arch}"

    def forward_features(self, x):
        """kan_vgg.py:142-146 `self.features(x)`, with one fusion: a B-spline or recurrence-family KAN layer directly followed by the "M" entry
        (MaxPool2d(2, 2)) runs the pooling inside its InstanceNorm+PReLU kernels, so the un-pooled activation and its
        gradient never go through HBM.  Same values as the two modules in sequence (tests/test_gpu_models.py)."""
        mods = list(self.features)
        i = 0
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if (self.fuse_pool and isinstance(m, (KANConvNDLayer, _RecurrenceKANConvNDLayer)) and getattr(m, "ndim", 2) == 2
                    and isinstance(nxt, nn.MaxPool2d) and _is_pool_2x2(nxt)):
                x = m(x, pool=True)
                i += 2
            else:
                x = m(x)
                i += 1
        return x

    def forward(self, x: torch.Tensor, **kwargs) -> torch.Tensor:
        x = self.avgpool(self.forward_features(x))
        return self.classifier(torch.flatten(x, 1))


def _is_pool_2x2(m: nn.MaxPool2d) -> bool:
    two = lambda v: v == 2 or v == (2, 2)
    zero = lambda v: v == 0 or v == (0, 0)
    one = lambda v: v == 1 or v == (1, 1)
    return two(m.kernel_size) and two(m.stride) and zero(m.padding) and one(m.dilation) and not m.ceil_mode and not m.return_indices


def vggkan(input_channels: int, num_classes: int, conv_type: str = "kanconv", kan_conv: Optional[str] = "KAN",
           classifier_type: str = "Linear", arch: str = "VGG16", **kwargs: Any) -> VGGKAN:
    """Factory with the reference's leading arguments (kan_vgg.py:307-343)."""
    if conv_type != "kanconv":
        raise NotImplementedError("only conv_type='kanconv' is on the accelerated path")
    return VGGKAN(input_channels, num_classes, kan_conv=kan_conv or "KAN", arch=arch, classifier_type=classifier_type, **kwargs)
