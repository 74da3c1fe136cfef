#!/usr/bin/env python3
"""Headline benchmark: images/sec, forward + loss + backward, KAN-VGG11 (B-spline conv, grid 5, order 3, SiLU,
InstanceNorm2d, Linear head), synthetic 3x32x32, batch 256 per GPU (BASELINE.json configs[2]; configs[3] for N>1).

  python bench.py --gpus N --steps K --warmup W          (N > 1: this process starts N ranks itself, see launch_ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = zero grads, forward, CrossEntropy loss, backward (+ bucketed RCCL all-reduce of the gradients for N>1);
no optimizer step.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# dmabuf IPC is the only mode the host driver supports: must be in the environment before the first HIP call of any rank
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# torch (whose import maps libamdhip64 / libhsa-runtime64 into the process) is loaded by the WORKERS only: the launcher parent of
# `--gpus N` decides from its arguments, counts GPUs in sysfs and starts its ranks without ever loading the HIP runtime -- on an 8-rank
# node there is no 9th process holding the devices open (tests/test_bench_launcher.py checks /proc/self/maps at Popen time).
torch = dist = F = None


def _load_torch():
    global torch, dist, F
    if torch is None:
        import torch as _torch
        import torch.distributed as _dist
        import torch.nn.functional as _F
        torch, dist, F = _torch, _dist, _F
    return torch

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
                          desc="one FastKANConv2DLayer(3, 64, 3) (RBF grid 8, SiLU, input InstanceNorm2d) on 256x3x32x32, fwd + bwd from an upstream gradient randn_like(y) "
                               "(BASELINE.json configs[1]; SURVEY.md 8(d))"),
}


def build_model(device, workload="kan_vgg11"):
    _load_torch()
    torch.manual_seed(0)
    if workload == "kan_vgg11":
        from convkan_amd.models import vggkan
        return vggkan(3, 10, arch="VGG11", kan_conv="KAN", classifier_type="Linear").to(device).train()
    if workload == "cheby_alexnet":
        from convkan_amd.models import alexnet_kan
        return alexnet_kan(num_classes=10, kan_conv="ChebyKAN", degree=4).to(device).train()
    import convkan_amd

    class OneLayer(torch.nn.Module):                       # y = layer(x), backward from a fixed upstream gradient randn_like(y)
        def __init__(self):
            super().__init__()
            self.layer = convkan_amd.FastKANConv2DLayer(3, 64, 3)
            self.upstream = None

        def forward(self, x):
            return self.layer(x)

        def step(self, x):
            y = self.layer(x)
            if self.upstream is None or self.upstream.shape != y.shape:
                self.upstream = torch.randn(y.shape, device=y.device, generator=torch.Generator(device=y.device).manual_seed(2))
            y.backward(self.upstream)
            return y.detach().flatten()[0]
    return OneLayer().to(device).train()


def one_step(model, x, t, reducer=None):
    model.zero_grad(set_to_none=True)
    if hasattr(model, "step"):                             # single-layer workload: its own step (no loss head)
        return model.step(x)
    loss = F.cross_entropy(model(x), t)
    loss.backward()
    if reducer is not None:
        reducer.finish()
    return loss


def traced_step(model, x, t, reducer):
    """One untimed step under reducer.trace_step: offsets (ms from the start of backward, this rank's HIP events) of every bucket's
    "last gradient written", collective start and end, the end of backward on the compute stream, and `exposed_ms` -- how long
    the compute stream then sits in finish() waiting for collectives (0 = the exchange hid completely under backward)."""
    out = {}
    try:
        model.zero_grad(set_to_none=True)
        loss = F.cross_entropy(model(x), t)
        with reducer.trace_step(out):
            out["t0"] = torch.cuda.Event(enable_timing=True)
            out["t0"].record(torch.cuda.current_stream())
            loss.backward()
            reducer.finish()
        out.pop("t0", None)
    except Exception as e:                                # diagnostics only
        out = {"trace_error": f"{type(e).__name__}: {e}"[:200]}
    return out


def train_step_timing(model, x, t, steps, warmup):
    """Auxiliary figure, N = 1 only: the same step followed by the fused AdamW update (generic_train.py:24: lr 1e-3,
    weight_decay 1e-4).  `value` above stays the fwd+bwd metric of BASELINE.json / SURVEY.md 8(d); this shows what a
    full training step costs.  Runs after the main measurement (it moves the parameters into the optimizer's flat block)."""
    _load_torch()
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


def csrc_sha16() -> str:
    """sha256 (first 16 hex digits) over the kernel sources the library is built from: the identity of the code a counter profile describes."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "convolutional-kan-for-image-classification_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        with open(os.path.join(d, fn), "rb") as f:
            h.update(fn.encode()); h.update(f.read())
    return h.hexdigest()[:16]


def profiled(kernel: str):
    """Committed counter evidence for a kernel family: HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes) and the
    matrix-pipe busy fraction (SQ_VALU_MFMA_BUSY_CYCLES pass).  A profile is used only if it records the csrc hash of the kernels that
    are built NOW (tools/pmc_report.py writes `csrc_sha16`): counters of an older kernel source are refused, not quoted --
    (None, "stale: ...", None) then."""
    fam = kernel.split("/")[0]
    traffic = busy = src = None
    now = csrc_sha16()
    for rnd in ("r03",):
        try:
            with open(os.path.join(ROOT, "profiles", f"{rnd}_hbm_traffic.json")) as f:
                doc = json.load(f)
            if doc.get("csrc_sha16") != now:
                src = f"stale: profiles/{rnd}_hbm_traffic.json was taken on csrc {doc.get('csrc_sha16')}, built now {now}"
                continue
            traffic = round(doc["kernels"][fam]["hbm_bytes_per_launch_corrected"])
            src = f"profiles/{rnd}_hbm_traffic.json"
            break
        except (OSError, KeyError, ValueError):
            continue
    try:
        with open(os.path.join(ROOT, "profiles", "r03_mfma_busy.json")) as f:
            doc = json.load(f)
        if doc.get("csrc_sha16") == now:
            busy = doc["kernels"][fam]
    except (OSError, KeyError, ValueError):
        pass
    return traffic, src, busy


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


def cpu_baseline(batch: int, iters: int, warmup: int):
    """the oracle (CPU restatement of the reference op sequence kan_layers.py:197-247, torch CPU ops) on
    this box's host cores -- batch 256, >= 3 warm-up + >= 10 timed iterations at every core this process may use, plus one short
    run pinned to 8 threads for comparison with the build container's indicative 28-33 img/s."""
    _load_torch()
    from oracle.kan_oracle import OracleKANVGG
    torch.manual_seed(0)
    m = OracleKANVGG().train()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(batch, 3, 32, 32, generator=g)
    t = torch.randint(0, 10, (batch,), generator=g)

    def run(threads, n_warm, n_iter):
        torch.set_num_threads(threads)
        times = []
        for _ in range(n_warm + n_iter):
            t0 = time.perf_counter()
            m.zero_grad(set_to_none=True)
            F.cross_entropy(m(x), t).backward()
            times.append(time.perf_counter() - t0)
        timed = times[n_warm:]
        return batch * len(timed) / sum(timed), min(timed), max(timed)

    threads = host_cores()
    ips, lo, hi = run(threads, warmup, iters)
    out = {"value": round(ips, 2), "unit": "images/sec", "cores": threads, "kind": "port",
           "sample": f"oracle KAN-VGG11 fwd+CE loss+bwd, batch {batch}, {iters} timed iterations after {warmup} warm-up "
                     f"(iteration {lo:.2f}-{hi:.2f} s), torch {torch.__version__} CPU, os.cpu_count()={os.cpu_count()}"}
    if threads != 8:
        n8 = max(2, iters // 3)
        ips8, lo8, hi8 = run(min(8, threads), 1, n8)
        out["threads_8"] = {"value": round(ips8, 2), "unit": "images/sec", "cores": min(8, threads),
                            "sample": f"same model and batch, torch.set_num_threads(8), {n8} timed iterations after 1 warm-up"}
    return out


# --------------------------------------------------------------------------------------------------- measurement
def summarise(prof, elapsed_s, steps, batch, world, gflop_per_image):
    """Per-kernel-family and per-layer roofline figures from the HIP-event samples of the timed steps."""
    fam, lay = {}, {}
    for smp in prof:
        ms = smp.start.elapsed_time(smp.end)
        f = fam.setdefault(smp.name, [0, 0.0, 0.0, 0.0])
        f[0] += 1; f[1] += ms; f[2] += smp.flops; f[3] += smp.executed
        l = lay.setdefault(smp.layer, {}).setdefault(smp.name.split("/")[0].replace("k_conv_", ""), [0, 0.0, 0.0, 0.0])
        l[0] += 1; l[1] += ms; l[2] += smp.flops; l[3] += smp.executed
    tf = lambda fl, ms: round(fl / (ms * 1e-3) / 1e12, 2)
    kernels = {n: {"launches": v[0], "avg_ms": round(v[1] / v[0], 4), "tflops": tf(v[2], v[1]), "executed_tflops": tf(v[3], v[1]),
                   "share_of_step": round(v[1] / (elapsed_s * 1e3), 3)} for n, v in fam.items()}
    layers = {ln: {k: {"avg_ms": round(v[1] / v[0], 4), "tflops": tf(v[2], v[1]), "executed_tflops": tf(v[3], v[1])} for k, v in ks.items()}
              for ln, ks in lay.items()}
    dom = max(fam, key=lambda n: fam[n][1])
    d = fam[dom]
    all_t, all_f, all_x = (sum(v[i] for v in fam.values()) for i in (1, 2, 3))
    ips = world * batch * steps / elapsed_s
    traffic, traffic_src, busy = profiled(dom)
    roof = {"bound": "mfma", "kernel": dom, "achieved": tf(d[2], d[1]), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(d[2] / (d[1] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "mfma_busy": busy,
            "avg_launch_ms": kernels[dom]["avg_ms"], "flops_per_launch": round(d[2] / d[0] / 1e9, 3),
            # work the matrix pipe really runs: position-major launches skip the (position, tap) products that multiply zero padding
            "executed_flops_per_launch": round(d[3] / d[0] / 1e9, 3),
            "executed_frac": round(d[3] / (d[1] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
            "all_conv_kernels_tflops": tf(all_f, all_t), "all_conv_kernels_executed_tflops": tf(all_x, all_t),
            "conv_kernel_share_of_step": round(all_t / (elapsed_s * 1e3), 3),
            # dense count per image (3 x forward, includes the first layer's bwd-data although no launch computes it)
            "end_to_end_frac": round(ips * gflop_per_image / 1e3 / world / FP32_MFMA_PEAK_TFLOPS, 4),
            # executed count: FLOPs of the launches that ran (no bwd-data of the first layer, no dead taps) / step time / peak
            "end_to_end_executed_frac": round(all_x / steps / (elapsed_s / steps) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
            "kernels": kernels, "layers": layers}
    return ips, roof


def timed_steps(model, x, t, steps, warmup, reducer=None, barrier=None):
    _load_torch()
    from convkan_amd import ops
    barrier = barrier or torch.cuda.synchronize
    for _ in range(warmup):
        one_step(model, x, t, reducer)
    barrier()
    ops.PROFILE = []                                  # HIP events around every conv-kernel launch of the timed steps
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = one_step(model, x, t, reducer)
    barrier()
    elapsed = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    return elapsed, prof, loss


def hip_graph_replay(model, x, t, iters: int = 100):
    """The same step captured once into a HIP graph (torch.cuda.graph: every launch of the step goes through ctypes onto torch's current stream, so the
    capture sees them all) and replayed: what the step costs without the host's per-launch work.  Only worth it where the step is launch-bound -- the
    single-layer workload (0.24 ms of kernels in a 0.48 ms eager step); on KAN-VGG11 the GPU is busy 100 % of the step.  The replayed step must reproduce
    the eager one bit for bit (output sample and every gradient), else the result is reported as not matching."""
    try:
        eager_y = float(one_step(model, x, t))
        eager_g = [p.grad.clone() for p in model.parameters() if p.grad is not None]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                one_step(model, x, t)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        model.zero_grad(set_to_none=True)
        with torch.cuda.graph(graph):
            y = one_step(model, x, t)
        for _ in range(5):
            graph.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(iters):
            graph.replay()
        e1.record(); torch.cuda.synchronize()
        same = float(y) == eager_y and all(torch.equal(a, p.grad) for a, p in zip(eager_g, [p for p in model.parameters() if p.grad is not None]))
        return {"ms_per_step": round(e0.elapsed_time(e1) / iters, 4), "replays": iters, "matches_eager_bitwise": bool(same)}
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def other_workload(name, device, steps, warmup):
    """Short leg for BASELINE.json configs[1] / configs[4] (N = 1 only; not the recorded metric): same step definition."""
    try:
        wl = WORKLOADS[name]
        model = build_model(device, name)
        g = torch.Generator(device=device).manual_seed(1)
        x = torch.randn(wl["batch"], *wl["shape"], device=device, generator=g)
        t = torch.randint(0, 10, (wl["batch"],), device=device, generator=g)
        elapsed, prof, _ = timed_steps(model, x, t, steps, warmup)
        ips, roof = summarise(prof, elapsed, steps, wl["batch"], 1, wl["gflop_per_image"])
        keep = ("kernel", "achieved", "frac", "executed_frac", "avg_launch_ms", "all_conv_kernels_tflops", "conv_kernel_share_of_step",
                "end_to_end_frac", "end_to_end_executed_frac", "kernels")
        out = {"workload": wl["desc"], "per_gpu_batch": wl["batch"], "steps": steps, "warmup": warmup,
               "ms_per_step": round(elapsed / steps * 1e3, 3), "images_per_sec": round(ips, 1), "roofline": {k: roof[k] for k in keep}}
        if name == "fastkan_layer":                       # launch-bound: the same step as one HIP graph
            out["hip_graph"] = hip_graph_replay(model, x, t)
            if "ms_per_step" in out["hip_graph"]:
                gips = wl["batch"] / out["hip_graph"]["ms_per_step"] * 1e3
                out["hip_graph"]["images_per_sec"] = round(gips, 1)
                out["hip_graph"]["end_to_end_frac"] = round(gips * wl["gflop_per_image"] / 1e3 / FP32_MFMA_PEAK_TFLOPS, 4)    # as roofline.end_to_end_frac, on the replay time
                out["hip_graph"]["end_to_end_executed_frac"] = round(roof["end_to_end_executed_frac"] * out["ms_per_step"] / out["hip_graph"]["ms_per_step"], 4)
        del model, x, t, prof
        torch.cuda.empty_cache()
        return out
    except Exception as e:                                # auxiliary: never take the headline measurement down with it
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def split_precision_forward(device, iters: int = 20):
    """OPT-IN mode, reported apart from everything else (`dtype` f32 above is the exact path and stays the default and the headline): the split-precision
    forward (3 x bf16 pieces, six bf16 MFMA products per k-block; kanconv.h kan_conv_fwd_split) of the three KAN-VGG11 layers in its scope next to the exact
    kernel of the same layer, same inputs; `max_diff_vs_exact` is max |z_split - z_exact| / max |z_exact| (tests/test_gpu_split.py holds the fp64 comparison).
    The top-level figures are the 256 -> 256 @ 8x8 layer."""
    import torch
    try:
        import convkan_amd as K
        from convkan_amd import ops

        def timed(fn):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(iters):
                fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters
        layers = []
        for C_, O_, hw in ((256, 256, 8), (128, 256, 8), (64, 128, 16)):
            torch.manual_seed(0)
            layer = K.KANConv2DLayer(C_, O_, 3, padding=1, base_activation=torch.nn.SiLU).to(device)      # as models/kan_vgg.py builds it
            x = torch.randn(256, C_, hw, hw, device=device)
            spec, wb, ws = layer.conv_spec(), layer.base_conv[0].weight.detach(), layer.spline_conv[0].weight.detach()
            z_s, wc = ops.kan_conv_fwd_split(spec, x, wb, ws)
            z_e = ops.kan_conv(spec, x, None, [wb], [ws])
            diff = float((z_s - z_e).abs().max() / z_e.abs().max())
            ms_s = timed(lambda: ops.kan_conv_fwd_split(spec, x, wb, ws, wc))
            ms_e = timed(lambda: ops.kan_conv(spec, x, None, [wb], [ws]))
            gf = 2.0 * 256 * hw * hw * O_ * C_ * 81 / 1e9
            layers.append({"layer": f"{C_}->{O_} @ {hw}x{hw}, bs 256", "split_ms": round(ms_s, 4), "exact_ms": round(ms_e, 4),
                           "split_tflops_fp32_equivalent": round(gf / ms_s, 1), "exact_tflops": round(gf / ms_e, 1), "speedup": round(ms_e / ms_s, 2),
                           "max_diff_vs_exact": diff})
            del layer, x, z_s, z_e, wc
        top = dict(layers[0])
        top.pop("layer")
        return dict({"workload": "conv stage forward of KANConv2DLayer 256->256 @ 8x8, bs 256 (opt-in split-precision mode vs the exact fp32 kernel incl. its slab sum)"},
                    **top, default=False, layers=layers)
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def inference_modes(device, iters: int = 30):
    """KAN-VGG11 forward only (eval, torch.no_grad, bs 256): the exact path, and the OPT-IN `ops.split_precision_inference()` switch under which the three
    layers in scope of kan_conv_fwd_split run their conv stage in split precision (`logits_max_diff_vs_exact` = max |difference| / max |exact logits|)."""
    import torch
    try:
        from convkan_amd import ops
        model = build_model(device, "kan_vgg11").eval()
        x = torch.randn(256, 3, 32, 32, device=device, generator=torch.Generator(device=device).manual_seed(4))

        def timed():
            for _ in range(5):
                y = model(x)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(iters):
                y = model(x)
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters, y
        with torch.no_grad():
            ms_e, y_e = timed()
            with ops.split_precision_inference():
                ms_s, y_s = timed()
        out = {"workload": "KAN-VGG11 forward only (eval, no_grad), 3x32x32, bs 256",
               "exact": {"ms_per_batch": round(ms_e, 3), "images_per_sec": round(256 / ms_e * 1e3, 1)},
               "split_precision_opt_in": {"ms_per_batch": round(ms_s, 3), "images_per_sec": round(256 / ms_s * 1e3, 1),
                                          "logits_max_diff_vs_exact": float((y_s - y_e).abs().max() / y_e.abs().max()), "default": False}}
        del model, x
        torch.cuda.empty_cache()
        return out
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"[:300]}


# --------------------------------------------------------------------------------------------------- multi-rank launch
def kfd_gpu_count():
    """GPUs of this node as the kernel driver lists them (KFD topology nodes with SIMDs; CPUs have simd_count 0), cut to the
    *_VISIBLE_DEVICES selection if one is set.  Pure sysfs: no HIP / HSA call, so the caller's process never opens a device.
    None when the topology is not readable (then the ranks themselves fail loudly if a device is missing)."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    if not os.path.isdir("/sys/class/kfd"):            # no amdgpu compute driver on this machine at all
        return 0
    try:
        n = 0
        for node in os.listdir(base):
            with open(os.path.join(base, node, "properties")) as f:
                props = dict(ln.split()[:2] for ln in f if len(ln.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
    except (OSError, ValueError):
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        sel = os.environ.get(var)
        if sel is not None:
            n = min(n, len([v for v in sel.split(",") if v.strip() != ""]))
    return n


def launch_ranks(n: int, argv, script: str = None, check_devices: bool = True, pre_popen=None):
    """`python bench.py --gpus N` with N > 1 (or --spawn): start N fresh worker processes -- one per GPU, a
    torch.distributed.run child -- relay rank 0's JSON line and exit with the child's code.  This parent never touches the GPU:
    it has not imported torch (no HIP / HSA library is even mapped), counts devices in sysfs (kfd_gpu_count) and re-executes nothing;
    it decides from the arguments alone, before anything else runs."""
    have = kfd_gpu_count() if check_devices else None
    if have is not None and have < n:
        raise SystemExit(f"bench.py --gpus {n}: this node exposes {have} GPU(s)")
    if pre_popen is not None:                          # test hook: inspect the parent just before it starts the ranks
        pre_popen()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script or os.path.abspath(__file__)] + [a for a in argv if a != "--spawn"]
    print("[bench] launching", " ".join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=dict(os.environ, KAN_BENCH_LAUNCHED="1"), text=True)
    line = None
    for ln in proc.stdout:                             # rank 0 prints exactly one JSON line on stdout; anything else is noise
        if ln.lstrip().startswith("{") and '"metric"' in ln:
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if rc == 0 and line is not None:                   # a rank that died after rank 0 printed makes the whole run invalid
        print(line, flush=True)
    elif rc == 0:
        rc = 1
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=0, help="images per GPU (default: the workload's)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="kan_vgg11")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux", action="store_true", help="skip the auxiliary legs (with-optimizer timing, other workloads): clean kernel profiles")
    ap.add_argument("--cpu-batch", type=int, default=256)
    ap.add_argument("--cpu-iters", type=int, default=10)
    ap.add_argument("--cpu-warmup", type=int, default=3)
    ap.add_argument("--bucket-mb", type=int, default=96)
    ap.add_argument("--force-dp", action="store_true", help="run the bucketed all-reduce path even with one rank (path rehearsal)")
    ap.add_argument("--spawn", action="store_true", help="start the ranks through the launcher even for --gpus 1")
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if (args.gpus > 1 or args.spawn) and world_env is None:
        launch_ranks(args.gpus, sys.argv[1:])           # does not return
    _load_torch()                                       # workers (and the plain single-GPU run) only
    # stdout carries exactly ONE line (the JSON).  Native libraries print banners there (RCCL: "RCCL version : ..." at
    # communicator creation), so fd 1 points at stderr until the result is ready.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start bench.py directly (it launches its own ranks) or with "
                         f"torch.distributed.run --nproc-per-node {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs a ROCm device"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dp
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)          # nccl == RCCL on ROCm

    import convkan_amd
    convkan_amd.build_library()

    wl = WORKLOADS[args.workload]
    if args.batch <= 0:
        args.batch = wl["batch"]
    model = build_model(device, args.workload)
    reducer = None
    if use_dist:
        from convkan_amd.parallel import BucketedGradReducer
        with torch.no_grad():
            for p in model.parameters():                           # identical replicas (same seed); make it explicit
                dist.broadcast(p, 0)
        reducer = BucketedGradReducer(model.parameters(), bucket_bytes=args.bucket_mb << 20, always_reduce=args.force_dp)
    g = torch.Generator(device=device).manual_seed(1 + rank)
    x = torch.randn(args.batch, *wl["shape"], device=device, generator=g)
    t = torch.randint(0, 10, (args.batch,), device=device, generator=g)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    elapsed, prof, loss = timed_steps(model, x, t, args.steps, args.warmup, reducer, barrier)
    if rank == 0:
        print(f"[bench] {args.steps} steps in {elapsed:.3f}s on {world} GPU(s)", file=sys.stderr, flush=True)
    per_rank, trace = [elapsed], {}
    if use_dist:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        every = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(every, tt)                       # each rank's own clock over the same K steps (spread = load imbalance / stragglers)
        per_rank = [float(v.item()) for v in every]
        elapsed = max(per_rank)                          # the contract's MAX over ranks
        # one more, UNTIMED step with HIP events around every bucket's collective: where the all-reduce sits relative to backward
        trace = traced_step(model, x, t, reducer)
        barrier()

    if rank == 0:
        ips, roof = summarise(prof, elapsed, args.steps, args.batch, world, wl["gflop_per_image"])
        out = {
            "metric": wl["metric"],
            "value": round(ips, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (randn images, randint labels; seeded random-init weights)",
            "config": {"workload": wl["desc"], "per_gpu_batch": args.batch,
                       "global_batch": args.batch * world, "parallelism": f"dp{world}" if world > 1 else "single",
                       "loss": round(float(loss.detach()), 6),
                       # cadence of the device-side weight fingerprint for layers whose host stamps are unchanged (ops._PACK_VERIFY_EVERY: 1 = every call)
                       "pack_verify_every": __import__("convkan_amd").ops._PACK_VERIFY_EVERY},
            # ranks of the RCCL communicator the gradient all-reduce ran on (0: single process, no collective in the step)
            "rccl_ranks": dist.get_world_size() if use_dist else 0,
            "launcher": "bench.py -> torch.distributed.run" if os.environ.get("KAN_BENCH_LAUNCHED") else ("torch.distributed.run" if world_env else "none"),
            "roofline": roof,
        }
        del prof
        if reducer is not None:
            out["allreduce"] = {"buckets": len(reducer.buckets), "bucket_mb": args.bucket_mb,
                                "bytes_per_step": sum(b.flat.numel() * 4 for b in reducer.buckets), "op": "avg, side stream, reverse order"}
            # exposed_ms, backward_end_ms, per-bucket ready / start / end offsets (rank 0, one untimed step)
            out["allreduce"].update({("bucket_trace" if k == "buckets" else k): v for k, v in trace.items()})
            out["ms_per_step_ranks"] = {"min": round(min(per_rank) / args.steps * 1e3, 3), "max": round(max(per_rank) / args.steps * 1e3, 3),
                                        "all": [round(v / args.steps * 1e3, 3) for v in per_rank]}
        aux = world == 1 and not use_dist and args.workload == "kan_vgg11" and not args.no_aux
        if aux:
            out["with_optimizer"] = train_step_timing(model, x, t, args.steps, max(3, args.warmup // 2))
            del model, x, t
            torch.cuda.empty_cache()
            out["other_workloads"] = {"fastkan_layer": other_workload("fastkan_layer", device, 30, 10),
                                      "cheby_alexnet": other_workload("cheby_alexnet", device, 10, 3),
                                      "split_precision_forward": split_precision_forward(device),
                                      "kan_vgg11_inference": inference_modes(device)}
        if world == 1 and not args.no_cpu_baseline and args.workload == "kan_vgg11":
            out["cpu_baseline"] = cpu_baseline(args.cpu_batch, args.cpu_iters, args.cpu_warmup)
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
