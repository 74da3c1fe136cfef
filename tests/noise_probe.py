#!/usr/bin/env python3
"""Where does the HIP path's distance from the fp64 oracle come from?  (the transcendentals were the suspected cause;
tests/exact_ab.py shows they are not.)  For a layer shape this prints, max-normalised against the fp64 oracle:
  z      the conv stage alone (pre-norm sums): HIP, the fp32 oracle (oneDNN), the fp32 oracle with oneDNN off (ATen's native conv: another
         summation order of the SAME arithmetic);
  n      InstanceNorm in fp64 applied to each of those z (isolates the norm: any difference left is z's error amplified by 1/sigma of the plane);
  y      the layer output of each implementation.
Lives under tests/ because it uses the oracle as the checker.   python tests/noise_probe.py > gpurun_out/noise_probe.txt"""
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import convkan_amd as K
from oracle import kan_oracle as O


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max())


def l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())


def run(C, Oc, H, B, seed=0):
    torch.manual_seed(seed)
    layer = K.KANConv2DLayer(C, Oc, 3, padding=1, base_activation=nn.SiLU)
    x = torch.randn(B, C, H, H)

    def oracle(dt):
        pre = []
        y = O.kan_conv2d(x.to(dt), [layer.base_conv[0].weight.detach().to(dt)], [layer.spline_conv[0].weight.detach().to(dt)],
                         [layer.prelus[0].weight.detach().to(dt)], knots=layer.grid.to(dt), spline_order=3, act=F.silu, padding=1, pre_norm_out=pre)
        return pre[0].detach(), y.detach()
    with torch.no_grad():
        z64, y64 = oracle(torch.float64)
        z32, y32 = oracle(torch.float32)
        with torch.backends.mkldnn.flags(enabled=False):
            z32n, y32n = oracle(torch.float32)
        lg = layer.cuda()
        xg = x.cuda()
        zh = K.ops.kan_conv(lg.conv_spec(), xg, None, [lg.base_conv[0].weight], [lg.spline_conv[0].weight]).cpu()
        yh = lg(xg).cpu()
    norm = lambda z: F.instance_norm(z.double(), eps=1e-5)
    sig = z64.double().flatten(2).std(dim=2, unbiased=False)
    amp = float((z64.double().flatten(2).abs().amax(dim=2) / (sig + 1e-30)).median())
    print(f"{C:4d}->{Oc:<4d}@{H}x{H} B={B:<4d} median max|z|/sigma per plane {amp:6.1f}")
    for tag, z, y in (("HIP", zh, yh), ("oracle fp32 (oneDNN)", z32, y32), ("oracle fp32 (ATen native conv)", z32n, y32n)):
        print(f"    {tag:32s} z max {rel(z, z64):.2e} L2 {l2(z, z64):.2e} | fp64-norm(z) max {rel(norm(z), norm(z64)):.2e} L2 {l2(norm(z), norm(z64)):.2e} | y max {rel(y, y64):.2e} L2 {l2(y, y64):.2e}")


if __name__ == "__main__":
    for shape in [(3, 64, 32, 8), (64, 128, 16, 8), (256, 256, 8, 8), (512, 512, 4, 16), (512, 512, 2, 16), (512, 512, 2, 128), (128, 128, 2, 128)]:
        run(*shape)
