#!/usr/bin/env python3
"""Diagnostic: per ChebyKAN-AlexNet layer shape, one launch over B images against the sum / concatenation of chunked launches
(HIP against HIP: samples are independent, the weight gradient is additive).  usage: chunk_consistency.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import convkan_amd as K
from convkan_amd.layers import ChebyKANConv2DLayer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
SHAPES = [(3, 64, 224, 11, 4, 2), (64, 192, 27, 5, 1, 2), (192, 384, 13, 3, 1, 1), (384, 256, 13, 3, 1, 1), (256, 256, 13, 3, 1, 1)]
torch.manual_seed(0)
for (C, O, H, k, s, p) in SHAPES:
    layer = ChebyKANConv2DLayer(C, O, kernel_size=k, degree=4, stride=s, padding=p, affine=True).cuda().train()
    x = torch.randn(B, C, H, H, device="cuda")
    with torch.no_grad():
        Ho = layer(x[:1]).shape[-1]
    dy = torch.randn(B, O, Ho, Ho, device="cuda")

    def run(lo, hi):
        layer.zero_grad(set_to_none=True)
        xi = x[lo:hi].clone().requires_grad_(True)
        y = layer(xi)
        y.backward(dy[lo:hi])
        torch.cuda.synchronize()
        return y.detach().double(), xi.grad.double(), {n: q.grad.double().clone() for n, q in layer.named_parameters()}

    def chunks(step):
        ys, dxs, gs = [], [], None
        for lo in range(0, B, step):
            y, dx, g = run(lo, min(B, lo + step))
            ys.append(y); dxs.append(dx)
            gs = g if gs is None else {n: gs[n] + g[n] for n in g}
        return torch.cat(ys), torch.cat(dxs), gs

    ref = chunks(8)
    print(f"--- {C}->{O} @{H} k{k} s{s}: B = {B}")
    for step in (16, 32, 64, B):
        if step > B: continue
        got = chunks(step)
        ey = float((got[0] - ref[0]).abs().max() / ref[0].abs().max())
        ex = float((got[1] - ref[1]).abs().max() / ref[1].abs().max())
        eg = {n: float((got[2][n] - ref[2][n]).abs().max() / (ref[2][n].abs().max() + 1e-30)) for n in ref[2]}
        print(f"  launches of {step:3d}: y {ey:.2e}  dx {ex:.2e}  " + "  ".join(f"{n.split('.')[0]}.{n.split('.')[-1]} {v:.2e}" for n, v in eg.items()))
