#!/usr/bin/env python3
"""ms/step (fwd+loss+bwd, bs 256, 3x32x32) of KAN-VGG11 built with each registered conv-KAN family."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import convkan_amd as K
from convkan_amd.models import vggkan
names = sys.argv[1:] or [n for n in K.CONV_KAN_FACTORY if n != "conv"]
x = torch.randn(256, 3, 32, 32, device="cuda"); t = torch.randint(0, 10, (256,), device="cuda")
for name in names:
    torch.manual_seed(0)
    m = vggkan(3, 10, arch="VGG11", kan_conv=name).cuda().train()
    def step():
        m.zero_grad(set_to_none=True)
        F.cross_entropy(m(x), t).backward()
    for _ in range(3): step()
    wins = []
    for _ in range(3):                                   # three windows of 10 steps, the median: one allocator hiccup inside a window used to read as +25 %
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): step()
        torch.cuda.synchronize()
        wins.append((time.perf_counter() - t0) * 100)
    ms = sorted(wins)[1]
    planes = [mod.conv_spec() for mod in m.modules() if hasattr(mod, "conv_spec")][1]
    P = planes.n_basis + (planes.act != -1)
    gf = 703.9 / 9 * P * 3          # dense fwd+bwd GFLOP per step at P planes
    print(f"{name:14s} P={P:2d} {ms:8.2f} ms/step  {256 / ms * 1e3:8.0f} img/s  {gf / ms:6.1f} TF (dense)")
    del m
