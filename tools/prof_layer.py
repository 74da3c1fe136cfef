#!/usr/bin/env python3
"""Profiling target: run the three conv kernels of one KAN-VGG11 layer a few times (for rocprofv3 --pmc / --kernel-trace)."""
import os, sys
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import convkan_amd as K
from convkan_amd import ops
VGG11 = [(3, 64, 32), (64, 128, 16), (128, 256, 8), (256, 256, 8), (256, 512, 4), (512, 512, 4), (512, 512, 2), (512, 512, 2)]
li = int(sys.argv[1]) if len(sys.argv) > 1 else 3
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
C, O, H = VGG11[li]
layer = K.KANConv2DLayer(C, O, 3, padding=1, base_activation=nn.SiLU).cuda()
spec = layer.conv_spec()
x = torch.randn(256, C, H, H, device="cuda"); dz = torch.randn(256, O, H, H, device="cuda")
wb, ws = [layer.base_conv[0].weight.detach()], [layer.spline_conv[0].weight.detach()]
for _ in range(iters):
    zs, packed, *_ = ops._conv_forward(spec, x, None, wb, ws, True)
    ops._conv_backward(spec, x, None, packed, dz, li != 0, False, True)
torch.cuda.synchronize()
