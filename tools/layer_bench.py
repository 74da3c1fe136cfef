#!/usr/bin/env python3
"""Per-layer, per-kernel timing of the KAN-VGG11 conv stages (tuning harness; HIP events on the launch stream).

  python tools/layer_bench.py [--batch 256] [--iters 5] [--layers 0,1,...]
Prints ms and TFLOP/s (dense algorithmic count) for forward, bwd-data, bwd-weight, plus pack/unpack and
the InstanceNorm+PReLU kernels."""
import argparse
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import convkan_amd as K            # noqa: E402
from convkan_amd import ops        # noqa: E402

VGG11 = [(3, 64, 32), (64, 128, 16), (128, 256, 8), (256, 256, 8), (256, 512, 4), (512, 512, 4), (512, 512, 2), (512, 512, 2)]


def timed(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--layers", default="0,1,2,3,4,5,6,7")
    a = ap.parse_args()
    K.build_library()
    tot = {"fwd": 0.0, "bwd_data": 0.0, "bwd_weight": 0.0, "other": 0.0}
    totf = 0.0
    for li in [int(v) for v in a.layers.split(",")]:
        C, O, H = VGG11[li]
        layer = K.KANConv2DLayer(C, O, 3, padding=1, base_activation=nn.SiLU).cuda()
        spec = layer.conv_spec()
        x = torch.randn(a.batch, C, H, H, device="cuda")
        dz = torch.randn(a.batch, O, H, H, device="cuda")
        wb, ws = [layer.base_conv[0].weight.detach()], [layer.spline_conv[0].weight.detach()]
        geom, basis, plan = ops._plan_cached(spec, a.batch, C, H, H, O, C, O)
        fl = ops._conv_flops(geom, plan)
        zs, packed, *_ = ops._conv_forward(spec, x, None, wb, ws, True)
        ops.PROFILE = []
        for _ in range(a.iters):
            ops._conv_forward(spec, x, None, wb, ws)
            ops._conv_backward(spec, x, None, packed, dz, li != 0, False, True)
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
        t = {}
        for smp in prof:
            key = smp.name.split("/")[0].replace("k_conv_", "").replace("_halo", "").replace("_pmdma", "")
            t[key] = t.get(key, 0.0) + smp.start.elapsed_time(smp.end) / a.iters
        t_all_f = timed(lambda: ops._conv_forward(spec, x, None, wb, ws), a.iters)
        t_all_b = timed(lambda: ops._conv_backward(spec, x, None, packed, dz, li != 0, False, True), a.iters)
        y = layer(x.requires_grad_(li != 0))
        t_layer_f = timed(lambda: layer(x), a.iters)
        other = t_all_f + t_all_b - sum(t.values())
        line = f"L{li} {C:3d}->{O:3d}@{H:2d} splits f/d/w={plan.fwd_splits}/{plan.bwd_data_splits}/{plan.bwd_weight_splits} {fl / 1e9:7.2f} GF |"
        for k in ("fwd", "bwd_data", "bwd_weight"):
            if k in t:
                line += f" {k} {t[k]:7.3f} ms {fl / t[k] / 1e9:6.1f} TF |"
                tot[k] += t[k]
        line += f" pack/unpack/reduce {other:6.3f} ms | layer fwd incl. IN {t_layer_f:7.3f} ms"
        tot["other"] += other
        totf += fl
        print(line, flush=True)
    s = sum(tot.values())
    print("total ms:", {k: round(v, 3) for k, v in tot.items()}, "sum", round(s, 3), f"=> {3 * totf / s / 1e9:.1f} TF on 3x fwd flops")


if __name__ == "__main__":
    main()
