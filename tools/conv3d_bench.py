"""fwd+bwd time of the 3-D shims (one 2-D launch set per depth tap) on a few volume shapes.  python tools/conv3d_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import convkan_amd as K  # noqa: E402


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (B, C, O, D) in [(8, 16, 32, 16), (4, 64, 128, 16), (2, 128, 128, 8)]:
    torch.manual_seed(0)
    layer = K.KANConv3DLayer(C, O, 3, padding=1).cuda()
    x = torch.randn(B, C, D, D, D, device="cuda", requires_grad=True)

    def step():
        layer.zero_grad(set_to_none=True)
        layer(x).square().mean().backward()
    ms = timed(step)
    gflop = 3 * 2.0 * B * O * D ** 3 * C * 9 * 27 / 1e9
    print(f"KANConv3DLayer {C}->{O} k3 on {B}x{C}x{D}^3: {ms:7.3f} ms fwd+bwd, {gflop / ms:6.1f} TFLOP/s on the dense count", flush=True)
