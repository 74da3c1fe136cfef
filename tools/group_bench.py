#!/usr/bin/env python3
"""Time grouped / depthwise KAN conv layers (fwd+bwd), ms per iteration.  usage: group_bench.py [repo_root]"""
import sys, time
root = sys.argv[1] if len(sys.argv) > 1 else "."
sys.path.insert(0, root)
import torch
import convkan_amd as K

CASES = [("depthwise C=96 28x28 B=64", 96, 96, 96, 28, 64), ("depthwise C=384 14x14 B=64", 384, 384, 384, 14, 64),
         ("groups=4 256->256 14x14 B=64", 256, 256, 4, 14, 64), ("groups=2 128->128 16x16 B=128", 128, 128, 2, 16, 128),
         ("groups=1 256->256 8x8 B=256", 256, 256, 1, 8, 256)]
for name, C, O, G, HW, B in CASES:
    torch.manual_seed(0)
    layer = K.KANConv2DLayer(C, O, 3, groups=G, padding=1, base_activation=torch.nn.SiLU).cuda()
    x = torch.randn(B, C, HW, HW, device="cuda", requires_grad=True)
    for it in range(13):
        if it == 3:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        layer.zero_grad(set_to_none=True)
        y = layer(x)
        y.backward(torch.ones_like(y))
    torch.cuda.synchronize()
    print(f"{name:34s} {(time.perf_counter() - t0) * 100:.3f} ms/iter  (y mean {float(y.mean()):+.5f})")
