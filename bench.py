#!/usr/bin/env python3
"""Headline benchmark: images/sec, forward + loss + backward, KAN-VGG11 (B-spline conv, grid 5, order 3, SiLU,
InstanceNorm2d, Linear head), synthetic 3x32x32, batch 256 per GPU (BASELINE.json configs[2]; configs[3] for N>1).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = zero grads, forward, CrossEntropy loss, backward (+ bucketed RCCL all-reduce of the gradients for N>1);
no optimizer step.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, exact fp32
GFLOP_PER_IMAGE = 8.249            # dense algorithmic count fwd+bwd for KAN-VGG11 @32x32


# Workloads: the headline (BASELINE.json configs[2]/[3]) is the default; the two other GPU configs of BASELINE.json can be
# timed with --workload (same step definition; they are not the metric the driver records).
WORKLOADS = {
    "kan_vgg11": dict(metric="images/sec fwd+bwd KAN-VGG11 3x32x32 bs=256/GPU", shape=(3, 32, 32), batch=256, gflop_per_image=8.249,
                      desc="KAN-VGG11 (KANConv2DLayer x8, grid=5, order=3, SiLU, InstanceNorm2d+PReLU, Linear head), 3x32x32, "
                           "fwd+CE loss+bwd, no optimizer step"),
    "cheby_alexnet": dict(metric="images/sec fwd+bwd KAN-AlexNet ChebyKAN(deg 4) 3x224x224 bs=128/GPU", shape=(3, 224, 224), batch=128,
                          gflop_per_image=2517.4 / 128, desc="KAN-AlexNet, ChebyKANConv2DLayer x5 (degree 4, InstanceNorm2d affine), FC head, "
                                                             "3x224x224, fwd+CE loss+bwd (BASELINE.json configs[4]); FLOPs = conv stages only"),
    "fastkan_layer": dict(metric="images/sec fwd+bwd single FastKANConv2DLayer 3->64 k3 on 3x32x32 bs=256/GPU", shape=(3, 32, 32), batch=256,
                          gflop_per_image=3 * 2.0 * 64 * 30 * 30 * 3 * 9 * 9 / 1e9,
                          desc="one FastKANConv2DLayer(3, 64, 3) (RBF grid 8, SiLU, input InstanceNorm2d), fwd + sum-loss + bwd (BASELINE.json configs[1])"),
}


def build_model(device, workload="kan_vgg11"):
    torch.manual_seed(0)
    if workload == "kan_vgg11":
        from convkan_amd.models import vggkan
        return vggkan(3, 10, arch="VGG11", kan_conv="KAN", classifier_type="Linear").to(device).train()
    if workload == "cheby_alexnet":
        from convkan_amd.models import alexnet_kan
        return alexnet_kan(num_classes=10, kan_conv="ChebyKAN", degree=4).to(device).train()
    import convkan_amd

    class OneLayer(torch.nn.Module):                       # logits = spatial mean of the layer output (10 of its 64 channels)
        def __init__(self):
            super().__init__()
            self.layer = convkan_amd.FastKANConv2DLayer(3, 64, 3)

        def forward(self, x):
            return self.layer(x).mean(dim=(2, 3))[:, :10]
    return OneLayer().to(device).train()


def one_step(model, x, t, reducer=None):
    model.zero_grad(set_to_none=True)
    loss = F.cross_entropy(model(x), t)
    loss.backward()
    if reducer is not None:
        reducer.finish()
    return loss


def train_step_timing(model, x, t, steps, warmup):
    """Auxiliary figure, N = 1 only: the same step followed by the fused AdamW update (generic_train.py:24: lr 1e-3,
    weight_decay 1e-4).  `value` above stays the fwd+bwd metric of BASELINE.json / SURVEY.md 8(d); this shows what a
    full training step costs.  Runs after the main measurement (it moves the parameters into the optimizer's flat block)."""
    try:
        from convkan_amd import FusedAdamW
        opt = FusedAdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)

        def step():
            opt.zero_grad()
            F.cross_entropy(model(x), t).backward()
            opt.step()
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        return {"optimizer": "FusedAdamW(lr=1e-3, weight_decay=1e-4)", "ms_per_step": round(el / steps * 1e3, 3),
                "images_per_sec": round(x.shape[0] * steps / el, 1), "steps": steps}
    except Exception as e:                                # auxiliary: never take the headline measurement down with it
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def profiled_traffic(kernel: str):
    """HBM bytes per launch of a kernel family, from the committed PMC pass (profiles/r01_hbm_traffic.json); None if absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")) as f:
            doc = json.load(f)
        fam = kernel.split("/")[0]
        return round(doc["kernels"][fam]["hbm_bytes_per_launch_corrected"])
    except (OSError, KeyError, ValueError):
        return None


def host_cores() -> int:
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota when one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                      # cgroup v2: "<quota> <period>" or "max <period>"
            q, p = f.read().split()
            if q != "max":
                n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, p = int(f.read()), int(g.read())
                if q > 0:
                    n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(batch: int, iters: int):
    """The oracle (CPU restatement of the reference op sequence, torch CPU ops) timed on this box's host cores."""
    from oracle.kan_oracle import OracleKANVGG
    threads = host_cores()
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    m = OracleKANVGG().train()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(batch, 3, 32, 32, generator=g)
    t = torch.randint(0, 10, (batch,), generator=g)
    times = []
    for i in range(iters + 1):
        t0 = time.perf_counter()
        m.zero_grad(set_to_none=True)
        F.cross_entropy(m(x), t).backward()
        times.append(time.perf_counter() - t0)
    best = sum(times[1:]) / iters                      # first iteration is warm-up
    return {"value": round(batch / best, 2), "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"oracle KAN-VGG11 fwd+loss+bwd, batch {batch}, mean of {iters} iters after 1 warm-up, torch {torch.__version__} CPU"}


def main():
    # stdout carries exactly ONE line (the JSON).  Native libraries print banners there (RCCL: "RCCL version : ..." at
    # communicator creation), so fd 1 points at stderr until the result is ready.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=0, help="images per GPU (default: the workload's)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="kan_vgg11")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux", action="store_true", help="skip the auxiliary with-optimizer timing (clean kernel profiles)")
    ap.add_argument("--cpu-batch", type=int, default=32)
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--bucket-mb", type=int, default=96)
    ap.add_argument("--force-dp", action="store_true", help="run the bucketed all-reduce path even with one rank (path rehearsal)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    assert torch.cuda.is_available(), "bench.py needs a ROCm device"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dp
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)          # nccl == RCCL on ROCm

    import convkan_amd
    from convkan_amd import ops
    convkan_amd.build_library()

    wl = WORKLOADS[args.workload]
    if args.batch <= 0:
        args.batch = wl["batch"]
    model = build_model(device, args.workload)
    reducer = None
    if use_dist:
        from convkan_amd.parallel import BucketedGradReducer
        for p in model.parameters():                               # identical replicas (same seed); make it explicit
            dist.broadcast(p.data, 0)
        reducer = BucketedGradReducer(model.parameters(), bucket_bytes=args.bucket_mb << 20, always_reduce=args.force_dp)
    g = torch.Generator(device=device).manual_seed(1 + rank)
    x = torch.randn(args.batch, *wl["shape"], device=device, generator=g)
    t = torch.randint(0, 10, (args.batch,), device=device, generator=g)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step(model, x, t, reducer)
    barrier()
    ops.PROFILE = []                                  # HIP events around every conv-kernel launch of the timed steps
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step(model, x, t, reducer)
    barrier()
    elapsed = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    if rank == 0:
        print(f"[bench] {args.steps} steps in {elapsed:.3f}s on {world} GPU(s)", file=sys.stderr, flush=True)
    if use_dist:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        # ---- per-kernel roofline from the HIP events
        fam = {}
        for name, flops, e0, e1 in prof:
            ms = e0.elapsed_time(e1)
            f = fam.setdefault(name, [0, 0.0, 0.0])
            f[0] += 1; f[1] += ms; f[2] += flops
        kernels = {n: {"launches": v[0], "avg_ms": round(v[1] / v[0], 4), "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 2),
                       "share_of_step": round(v[1] / (elapsed * 1e3), 3)} for n, v in fam.items()}
        dom = max(fam, key=lambda n: fam[n][1])
        ach = fam[dom][2] / (fam[dom][1] * 1e-3) / 1e12
        all_t = sum(v[1] for v in fam.values())
        all_f = sum(v[2] for v in fam.values())
        ips = world * args.batch * args.steps / elapsed
        out = {
            "metric": wl["metric"],
            "value": round(ips, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (randn images, randint labels; seeded random-init weights)",
            "config": {"workload": wl["desc"], "per_gpu_batch": args.batch,
                       "global_batch": args.batch * world, "parallelism": f"dp{world}" if world > 1 else "single",
                       "loss": round(float(loss.detach()), 6)},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": profiled_traffic(dom),
                         "traffic_source": "profiles/r01_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; bytes/launch)",
                         "avg_launch_ms": kernels[dom]["avg_ms"], "flops_per_launch": round(fam[dom][2] / fam[dom][0] / 1e9, 3),
                         "all_conv_kernels_tflops": round(all_f / (all_t * 1e-3) / 1e12, 2),
                         "conv_kernel_share_of_step": round(all_t / (elapsed * 1e3), 3),
                         "end_to_end_frac": round(ips * wl["gflop_per_image"] / 1e3 / world / FP32_MFMA_PEAK_TFLOPS, 4),
                         "kernels": kernels},
        }
        if world == 1 and args.workload == "kan_vgg11" and not args.no_aux:
            out["with_optimizer"] = train_step_timing(model, x, t, args.steps, max(3, args.warmup // 2))
        if world == 1 and not args.no_cpu_baseline and args.workload == "kan_vgg11":
            out["cpu_baseline"] = cpu_baseline(args.cpu_batch, args.cpu_iters)
            out["cpu_baseline"]["gpu_over_cpu"] = round(ips / out["cpu_baseline"]["value"], 1)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)                                     # anything printed during teardown goes to stderr again
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
