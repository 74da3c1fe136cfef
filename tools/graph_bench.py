#!/usr/bin/env python3
"""Training step (zero_grad / fwd / CE loss / bwd / FusedAdamW) of KAN-VGG11: eager against train.GraphedStep (fwd + loss + bwd as one HIP graph, the
optimizer step outside it), over batch sizes.  usage: graph_bench.py [kan_conv] [batch ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import convkan_amd as K
from convkan_amd.models import vggkan
kind = sys.argv[1] if len(sys.argv) > 1 else "KAN"
batches = [int(v) for v in sys.argv[2:]] or [8, 32, 64, 256]
for B in batches:
    torch.manual_seed(0)
    model = vggkan(3, 10, arch="VGG11", kan_conv=kind).cuda().train()
    opt = K.FusedAdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
    x = torch.randn(B, 3, 32, 32, device="cuda"); t = torch.randint(0, 10, (B,), device="cuda")
    def timed(fn, n=30):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    eager = timed(lambda: K.train_step(model, x, t, opt))
    step = K.GraphedStep(model, x, t)
    def g():
        step(x, t); opt.step()
    graph = timed(g)
    print(f"{kind}-VGG11 bs {B:4d}: eager {eager:7.3f} ms/step  graphed {graph:7.3f} ms/step  ({eager / graph:.2f}x)", flush=True)
    del model, opt, step
    torch.cuda.empty_cache()
