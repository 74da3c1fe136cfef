#!/usr/bin/env python3
"""Per-layer check at the headline batch: every KAN-VGG11 layer shape at B = 256, HIP fwd+bwd against the CPU oracle run in
fp64 and in fp32 (L2-relative and max-normalised errors of y, dx and each parameter gradient).  GPU box only."""
import copy, os, sys, time
import torch, torch.nn as nn
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import convkan_amd as K
from helpers import oracle_forward

VGG11 = [(3, 64, 32), (64, 128, 16), (128, 256, 8), (256, 256, 8), (256, 512, 4), (512, 512, 4), (512, 512, 2), (512, 512, 2)]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
layers = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else range(8)
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
l2 = lambda a, b: float((a.double().cpu() - b.double()).norm() / (b.double().norm() + 1e-300))
mx = lambda a, b: float((a.double().cpu() - b.double()).abs().max() / (b.double().abs().max() + 1e-300))
for li in layers:
    C, O, H = VGG11[li]
    torch.manual_seed(li)
    layer = K.KANConv2DLayer(C, O, 3, padding=1, base_activation=nn.SiLU)
    cfg = dict(kind="bspline", C=C, O=O, k=3, s=1, p=1, d=1, groups=1, act="silu")
    x_cpu = torch.randn(B, C, H, H) * scale
    g = torch.Generator().manual_seed(99)
    go = torch.randn(B, O, H, H, generator=g)
    if os.environ.get("PLANE_CONST_DY"):                 # gradient of a spatial mean: constant over each plane (the last VGG layer)
        go = torch.randn(B, O, 1, 1, generator=g).expand(B, O, H, H).contiguous()
    if os.environ.get("POS_INPUT"):                      # post-InstanceNorm+PReLU-like input statistics
        x_cpu = torch.nn.functional.prelu(torch.nn.functional.instance_norm(x_cpu), torch.tensor([0.25]))
    res = {}
    for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
        lay = copy.deepcopy(layer).to(dt)
        xo = x_cpu.clone().to(dt).requires_grad_(True)
        t0 = time.time()
        yo = oracle_forward(cfg, lay, xo)
        yo.backward(go.to(dt))
        res[tag] = dict(y=yo.detach(), dx=xo.grad, **{n: p.grad for n, p in lay.named_parameters()})
        res[tag + "_s"] = time.time() - t0
    dev = copy.deepcopy(layer).cuda()
    x = x_cpu.clone().cuda().requires_grad_(True)
    y = dev(x)
    y.backward(go.cuda())
    torch.cuda.synchronize()
    got = dict(y=y.detach(), dx=x.grad, **{n: p.grad for n, p in dev.named_parameters()})
    print(f"L{li} {C}->{O}@{H} B={B}  (oracle {res['f64_s']:.1f}s / {res['f32_s']:.1f}s)")
    for k in got:
        print(f"   {k:26s} HIP-vs-f64 l2 {l2(got[k], res['f64'][k]):.2e} max {mx(got[k], res['f64'][k]):.2e} | "
              f"oracle f32-vs-f64 l2 {l2(res['f32'][k], res['f64'][k]):.2e} max {mx(res['f32'][k], res['f64'][k]):.2e}", flush=True)
