#!/usr/bin/env python3
"""Timing of BASELINE.json configs[1] (single FastKANConv2DLayer 3->64 on 256x3x32x32) and configs[4] (KAN-AlexNet,
ChebyKAN degree 4, 3x224x224, bs=128): fwd+bwd, HIP events."""
import os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import convkan_amd as K
from convkan_amd import ops
from convkan_amd.models import alexnet_kan

def timed(fn, iters=5):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

K.build_library()
torch.manual_seed(0)
layer = K.FastKANConv2DLayer(3, 64, 3).cuda()
x = torch.randn(256, 3, 32, 32, device="cuda", requires_grad=True)
g = torch.randn(256, 64, 30, 30, device="cuda")
def step2():
    layer.zero_grad(set_to_none=True); x.grad = None
    layer(x).backward(g)
ms = timed(step2, 20)
fl = 3 * 2.0 * 256 * 64 * 30 * 30 * 3 * 9 * 9
print(f"config2 FastKANConv2DLayer 3->64 k3 256x3x32x32 fwd+bwd: {ms:.3f} ms  ({fl / ms / 1e9:.1f} TF dense-count, {256 / ms * 1e3:.0f} img/s)")

m = alexnet_kan(num_classes=10, kan_conv="ChebyKAN", degree=4).cuda().train()
xa = torch.randn(128, 3, 224, 224, device="cuda"); t = torch.randint(0, 10, (128,), device="cuda")
def step5():
    m.zero_grad(set_to_none=True)
    F.cross_entropy(m(xa), t).backward()
ops.PROFILE = None
ms = timed(step5, 5)
print(f"config5 KAN-AlexNet ChebyKAN deg4 bs=128 3x224x224 fwd+bwd: {ms:.2f} ms  ({128 / ms * 1e3:.0f} img/s, {2517.4 / ms:.1f} TF on the 2517.4 GFLOP dense conv count)")
ops.PROFILE = []
step5(); torch.cuda.synchronize()
fam = {}
for smp in ops.PROFILE:
    f = fam.setdefault(smp.name, [0, 0.0, 0.0]); f[0] += 1; f[1] += smp.start.elapsed_time(smp.end); f[2] += smp.flops
for n, v in fam.items(): print(f"   {n:28s} launches {v[0]:2d} total {v[1]:7.2f} ms  {v[2] / v[1] / 1e9:6.1f} TF")
