#!/usr/bin/env python3
"""A/B of the hot loops' transcendentals: the shipped library (hardware v_exp_f32 / v_rcp_f32 SiLU, RBF, tanh; closed-form
fp32 B-spline pieces) against a -DKAN_EXACT_TRANSCENDENTALS build of the SAME kernels (libm expf / tanhf, IEEE division, B-spline pieces in
double) -- does the HIP path's distance from the fp64 oracle fall to the fp32 oracle's, and what does it cost?

    python tests/exact_ab.py [--out profiles/r03_exact_transcendentals_ab.json]          (GPU box; ~4 min incl. the variant build)

Per variant, in a fresh process (the library is loaded once per process):
  layers   the eight KAN-VGG11 layer shapes (B = 8; 16 on the 2x2 planes), fwd + bwd: max-normalised error of y / dx / dW against the fp64
           oracle next to the fp32 oracle's own (helpers.check_vs_oracle, nothing asserted here);
  model    KAN-VGG11 bs 256 forward: every layer's input activation and the logits against the fp64 oracle model (L2-relative), next to the
           fp32 oracle model's; then ms/step of fwd + loss + bwd (10 warm-up + 30 timed).
This file lives under tests/ because it uses the oracle as the checker (only tests/, smoke() and bench.py's cpu_baseline may touch oracle/)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
DEFINE = "KAN_EXACT_TRANSCENDENTALS"


def measure(variant: str) -> dict:
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    import convkan_amd as K
    from convkan_amd import _lib as L
    from convkan_amd import build
    if variant == "exact":
        L.LIB_PATH = build.build_library(defines=(DEFINE,))
    else:
        K.build_library()
    from helpers import check_vs_oracle
    from test_gpu_oracle import VGG11, _cfg
    out = {"library": os.path.basename(L.LIB_PATH), "layers": {}, "model": {}}
    for li, (C, O, H) in enumerate(VGG11):
        torch.manual_seed(li)
        layer = K.KANConv2DLayer(C, O, 3, padding=1, base_activation=nn.SiLU)
        errs = check_vs_oracle(layer, _cfg("bspline", C, O, act="silu"), torch.randn(8 if H > 2 else 16, C, H, H), assert_ok=False)
        short = {"y": "y", "dx": "dx", "base_conv.0.weight": "dW_base", "spline_conv.0.weight": "dW_spline", "prelus": "d_prelu"}
        out["layers"][f"{C}->{O}@{H}x{H}"] = {short[k]: {"hip_vs_fp64": float(f"{v[0]:.3e}"), "oracle_fp32_vs_fp64": float(f"{v[2]:.3e}")} for k, v in errs.items()}
    # ---- bs 256 model: forward activations (what the PReLU gates downstream amplify), then step time
    import copy
    from convkan_amd.models import vggkan
    from oracle.kan_oracle import OracleKANConv2d, OracleKANVGG
    from test_gpu_models import _capture
    torch.manual_seed(0)
    m = vggkan(3, 10, arch="VGG11", kan_conv="KAN", classifier_type="Linear", dropout_linear=0.0)
    m.fuse_pool = False
    o = OracleKANVGG()
    o.classifier[0].p = 0.0
    o.load_state_dict({k: v.clone() for k, v in zip(o.state_dict().keys(), m.state_dict().values())})
    o64 = copy.deepcopy(o).double()
    for q in o64.modules():
        if isinstance(getattr(q, "knots", None), torch.Tensor):
            q.knots = q.knots.double()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(256, 3, 32, 32, generator=g)
    t = torch.randint(0, 10, (256,), generator=g)
    rec32, l32, _ = _capture(o, OracleKANConv2d, x, t)
    rec64, l64, _ = _capture(o64, OracleKANConv2d, x.double(), t)
    m = m.cuda()
    rech, lh, _ = _capture(m, K.KANConvNDLayer, x.cuda(), t.cuda())
    dist = lambda a, b: float((a.double().cpu() - b.double().cpu()).norm() / (b.double().norm() + 1e-300))
    acts = {}
    for i in sorted(rech):
        acts[f"features.{i}.input"] = {"hip_vs_fp64": float(f"{dist(rech[i]['x'], rec64[i]['x']):.3e}"), "oracle_fp32_vs_fp64": float(f"{dist(rec32[i]['x'], rec64[i]['x']):.3e}")}
    acts["logits"] = {"hip_vs_fp64": float(f"{dist(lh, l64):.3e}"), "oracle_fp32_vs_fp64": float(f"{dist(l32, l64):.3e}")}
    out["model"]["activations_L2"] = acts
    m.fuse_pool = True
    xd, td = x.cuda(), t.cuda()

    def step():
        m.zero_grad(set_to_none=True)
        F.cross_entropy(m(xd), td).backward()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    out["model"]["ms_per_step"] = round((time.perf_counter() - t0) / 30 * 1e3, 3)
    return out


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":
        print("@@RESULT@@" + json.dumps(measure(sys.argv[2])), flush=True)
        return
    out_path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(ROOT, "gpurun_out", "exact_ab.json")
    doc = {"note": "shipped library (hardware exp2 / rcp transcendentals, closed-form fp32 B-spline pieces) vs a -DKAN_EXACT_TRANSCENDENTALS build of the same kernels "
                   "(libm expf / tanhf, IEEE division, B-spline pieces in double); errors max-normalised (layers) / L2-relative (model) against the fp64 oracle, "
                   "next to the fp32 oracle's own distance from fp64; tests/exact_ab.py", "variants": {}}
    for variant in ("fast", "exact"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--variant", variant], capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("@@RESULT@@")]
        if r.returncode != 0 or not line:
            sys.stderr.write(r.stderr[-3000:])
            raise SystemExit(f"variant {variant} failed (rc {r.returncode})")
        doc["variants"][variant] = json.loads(line[0][len("@@RESULT@@"):])
        print(f"[{variant}] {doc['variants'][variant]['model']['ms_per_step']} ms/step", flush=True)
    f, e = doc["variants"]["fast"], doc["variants"]["exact"]
    worst = lambda v, key: max(d[key] / max(d["oracle_fp32_vs_fp64"], 1e-30) for d in v["model"]["activations_L2"].values())
    doc["summary"] = {"ms_per_step": {"fast": f["model"]["ms_per_step"], "exact": e["model"]["ms_per_step"]},
                      "worst_activation_error_over_fp32_oracle": {"fast": round(worst(f, "hip_vs_fp64"), 2), "exact": round(worst(e, "hip_vs_fp64"), 2)}}
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    with open(out_path, "w") as fh:
        json.dump(doc, fh, indent=1)
    print(json.dumps(doc["summary"]))


if __name__ == "__main__":
    main()
