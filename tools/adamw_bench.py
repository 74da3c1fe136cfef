"""HBM roofline of the fused AdamW step (28 B per element) and the cost of a full training step with it on KAN-VGG11 bs 256,
next to torch.optim.AdamW (foreach and fused) on the same model.  python tools/adamw_bench.py  [--json out.json]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import convkan_amd as K  # noqa: E402
from convkan_amd.models import vggkan  # noqa: E402

PEAK_GBS = 8000.0


def timed(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    out = {}
    torch.manual_seed(0)
    x, t = torch.randn(256, 3, 32, 32, device="cuda"), torch.randint(0, 10, (256,), device="cuda")
    crit = torch.nn.CrossEntropyLoss()
    for name in ("fused_flat", "torch_foreach", "torch_fused"):
        torch.manual_seed(0)
        m = vggkan(3, 10, arch="VGG11", kan_conv="KAN", classifier_type="Linear").cuda().train()
        nparam = sum(p.numel() for p in m.parameters())
        if name == "fused_flat":
            opt = K.FusedAdamW(m.parameters(), lr=1e-3, weight_decay=1e-4)
        else:
            opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-4, foreach=name == "torch_foreach", fused=name == "torch_fused")
        K.train_step(m, x, t, opt, crit)                       # gradients exist from here on

        def fwd_bwd():
            opt.zero_grad()
            crit(m(x), t).backward()
        ms_step_only = timed(opt.step)
        ms_fb = timed(fwd_bwd, n=10, warm=3)
        ms_full = timed(lambda: K.train_step(m, x, t, opt, crit), n=10, warm=3)
        gbs = 28.0 * nparam / (ms_step_only * 1e-3) / 1e9
        out[name] = dict(params=nparam, optimizer_step_ms=round(ms_step_only, 4), algorithmic_GBps=round(gbs, 1),
                         frac_of_hbm_peak=round(gbs / PEAK_GBS, 3), fwd_bwd_ms=round(ms_fb, 3), train_step_ms=round(ms_full, 3),
                         images_per_s=round(256 / (ms_full * 1e-3), 1))
        print(name, json.dumps(out[name]), flush=True)
        del m, opt
        torch.cuda.empty_cache()
    for a in sys.argv:
        if a.startswith("--json="):
            json.dump(out, open(a.split("=", 1)[1], "w"), indent=1)


if __name__ == "__main__":
    t0 = time.time()
    main()
    print(f"done in {time.time() - t0:.0f} s")
