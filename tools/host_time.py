#!/usr/bin/env python3
"""Host-side enqueue time vs GPU time of one KAN-VGG11 step (is the step launch-bound?)."""
import sys, time
sys.path.insert(0, ".")
import torch, torch.nn.functional as F
from convkan_amd.models import vggkan
torch.manual_seed(0)
m = vggkan(3, 10, arch="VGG11", kan_conv="KAN").cuda().train()
x = torch.randn(256, 3, 32, 32, device="cuda"); t = torch.randint(0, 10, (256,), device="cuda")
def step():
    m.zero_grad(set_to_none=True)
    F.cross_entropy(m(x), t).backward()
for _ in range(5): step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"rep {rep}: host enqueue {1e3*(t1-t0)/20:.2f} ms/step, total {1e3*(t2-t0)/20:.2f} ms/step")
