"""Model-level GPU parity: KAN-VGG11 and ChebyKAN-AlexNet built from this repo's layers vs. the reference models'
logits / loss / per-parameter gradient norms frozen in tests/golden/model_*.npz (parameters are set by the same
machine-independent fill the generator used, in named_parameters order)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

# Model-level fp32 tolerances are CALIBRATED, not chosen: tests/golden/make_golden.py (--model-only) stores, next to the
# reference's fp32 pass, its fp64 pass and how far the reference itself moves at model level
#   * fp32 vs its own fp64 pass under three executions of the SAME ops (default, 1 thread, oneDNN off = another summation order),
#   * those fp32 executions against each other, and
#   * the exact (fp64) model's response to one fp32 rounding (2^-24 relative) of its parameters and input.
# KAN-VGG11: with oneDNN off the reference moves by 1.7e-4 on the logits and 5.9e-4 on gradient norms against its own default
# pass; ChebyKAN-AlexNet: 3.6e-3 / 1.0e-2 on gradient norms / slices against fp64 for EVERY variant (acos near the clamp).
# A correct fp32 implementation with its own summation order therefore cannot be asked for less than ~that spread; the test
# asserts  err <= max(stated SURVEY 8(c) tolerance, K_SPREAD x reference spread)  against BOTH the fp32 and the fp64 reference pass.
K_SPREAD = 4.0
STATED = {"logits": 1e-4, "loss": 1e-4, "grad_norm": 1e-3, "grad_slice": 1e-3}      # SURVEY.md section 8(c), model level


def calibrated_tolerances(d):
    cal = json.loads(bytes(d["calib"]).decode())
    spread = {k: max([v[k] for v in cal["fp32_vs_fp64"].values()] + [v[k] for v in cal["fp32_vs_fp32"].values()] + [cal["eps_response"][k]])
              for k in STATED}
    return {k: max(STATED[k], K_SPREAD * spread[k]) for k in STATED}, spread


def det_fill(t, salt, scale):
    i = torch.arange(t.numel(), dtype=torch.float64)
    t.copy_((scale * torch.sin(i * 0.6180339887498949 * 7.0 + salt * 1.2345 + 0.1)).to(torch.float32).view_as(t))


def model_fill(model):
    with torch.no_grad():
        for j, (n, p) in enumerate(model.named_parameters()):
            if p.dim() == 4:
                det_fill(p, j, (3.0 / (p.shape[1] * p.shape[2] * p.shape[3])) ** 0.5)
            elif p.dim() == 2:
                det_fill(p, j, (1.0 / p.shape[1]) ** 0.5)
            elif "prelus" in n:
                p.fill_(0.25)
            elif n.endswith("bias"):
                det_fill(p, j, 0.05)
            else:
                det_fill(p, j, 0.2); p.add_(1.0)


def run(name, model):
    d = np.load(os.path.join(GOLDEN, f"model_{name}.npz"))
    names = json.loads(bytes(d["names"]).decode())
    assert names == [n for n, _ in model.named_parameters()]
    tol, spread = calibrated_tolerances(d)
    model_fill(model)                       # on CPU, exactly as the generator did
    model = model.cuda().eval()             # eval: head Dropout inert (the generator did the same)
    x = torch.from_numpy(d["x"]).cuda()
    t = torch.from_numpy(d["t"]).cuda()
    logits = model(x)
    loss = F.cross_entropy(logits, t)
    loss.backward()
    torch.cuda.synchronize()
    got_logits = logits.detach().double().cpu().numpy()
    gn = np.array([float(p.grad.double().norm()) for _, p in model.named_parameters()])
    head = np.stack([np.pad(p.grad.flatten()[:64].double().cpu().numpy(), (0, max(0, 64 - p.numel()))) for _, p in model.named_parameters()])
    for tag, sfx in (("fp32 reference", ""), ("fp64 reference", "64")):
        ref_logits = d["logits" + sfx].astype(np.float64)
        err = float(np.abs(got_logits - ref_logits).max() / np.abs(ref_logits).max())
        lerr = abs(float(loss) - float(d["loss" + sfx])) / max(1.0, abs(float(d["loss" + sfx])))
        rel = np.abs(gn - d["grad_norm" + sfx]) / (d["grad_norm" + sfx] + 1e-30)
        herr = np.abs(head - d["grad_head" + sfx]).max(axis=1) / (d["grad_absmax" + sfx] + 1e-30)
        print(f"[{name} vs {tag}] logits {err:.2e} (tol {tol['logits']:.1e}, ref spread {spread['logits']:.1e})  loss {lerr:.2e}  "
              f"grad-norm {rel.max():.2e} at {names[int(rel.argmax())]} (tol {tol['grad_norm']:.1e}, spread {spread['grad_norm']:.1e})  "
              f"grad-slice {herr.max():.2e} at {names[int(herr.argmax())]} (tol {tol['grad_slice']:.1e}, spread {spread['grad_slice']:.1e})")
        assert err <= tol["logits"], f"{tag}: logits err {err:.3e} > {tol['logits']:.3e}"
        assert lerr <= tol["loss"], f"{tag}: loss err {lerr:.3e}"
        assert rel.max() <= tol["grad_norm"], f"{tag}: grad-norm rel err {rel.max():.3e} at {names[int(rel.argmax())]}"
        assert herr.max() <= tol["grad_slice"], f"{tag}: grad slice err {herr.max():.3e} at {names[int(herr.argmax())]}"


def test_kan_vgg11(gpu_lib):
    from convkan_amd.models import vggkan
    torch.manual_seed(0)
    run("kan_vgg11", vggkan(3, 10, arch="VGG11", kan_conv="KAN", classifier_type="Linear"))


def test_cheby_alexnet(gpu_lib):
    from convkan_amd.models import alexnet_kan
    torch.manual_seed(0)
    run("cheby_alexnet", alexnet_kan(num_classes=10, kan_conv="ChebyKAN", degree=4))


def test_cheby_alexnet_config5_full_batch(gpu_lib):
    """BASELINE.json configs[4] at its FULL size (ChebyKAN-AlexNet, 128 x 3x224x224) -- too big for the CPU oracle, so it is tied to the
    reference through size-independent properties of the path (samples are independent through conv, per-sample InstanceNorm and PReLU;
    the weight gradient is additive over samples):
      (a) sample 0 of the batch IS the reference fixture's image: its logits inside the 128-image launch match the reference's fp32 and
          fp64 passes at the calibrated model-level tolerance, and every image's logits equal those of chunked launches (1 + 31 + 32 + 64);
      (b) every conv-KAN layer, on the input and the upstream gradient it actually saw inside the 128-image step, is re-run alone in chunks
          of 32 images (other tile counts, split-K factors and kernel choices): outputs and input gradients of the full launch equal the
          chunks' and every parameter gradient equals the sum over the chunks, at 2 x the stated layer tolerances (two launches, each within
          the stated tolerance of the exact result); the layers that carry the fused MaxPool2d(3, 2) assert values against the chunks and
          gradients against the batch in reversed order (same launch configuration, hence the same window choices);
      (c) the 1-image chunk of the model is the fixture's own step: its gradients match the reference's.
    Model-level gradients of the full launch against the sum of chunked MODEL steps are printed, not asserted: MaxPool2d(3, 2) argmax and
    PReLU decisions re-route under the 1e-6 forward differences between launch configurations (measured round 3: logits agree to 1e-6,
    gradients differ by up to 2e-1 max-normalised while every layer alone agrees to 2e-6 -- tools/probe/chunk_consistency.py)."""
    from convkan_amd.models import alexnet_kan
    from convkan_amd.layers import ChebyKANConv2DLayer
    d = np.load(os.path.join(GOLDEN, "model_cheby_alexnet.npz"))
    tol, spread = calibrated_tolerances(d)
    torch.manual_seed(0)
    model = alexnet_kan(num_classes=10, kan_conv="ChebyKAN", degree=4)
    model_fill(model)
    model = model.cuda().eval()                              # eval: the head's Dropout is inert, the conv path is the same in both modes
    names = [n for n, _ in model.named_parameters()]
    g = torch.Generator().manual_seed(5)
    x = torch.randn(128, 3, 224, 224, generator=g)
    x[0] = torch.from_numpy(d["x"])[0]
    t = torch.randint(0, 10, (128,), generator=g)
    t[0] = int(d["t"][0])
    x, t = x.cuda(), t.cuda()

    def step(lo, hi):
        model.zero_grad(set_to_none=True)
        xi = x[lo:hi].clone().requires_grad_(True)
        logits = model(xi)
        F.cross_entropy(logits, t[lo:hi], reduction="sum").backward()
        torch.cuda.synchronize()
        return logits.detach().double(), xi.grad.double(), [p.grad.double().clone() for _, p in model.named_parameters()]

    # the full step, with every conv-KAN layer's input, output and both gradients recorded
    rec, hooks = {}, []
    for i, f in enumerate(model.features):
        if isinstance(f, ChebyKANConv2DLayer):
            def pre(mod, args, kwargs, i=i):
                rec.setdefault(i, {})["x"] = args[0].detach()
                rec[i]["pool"] = kwargs.get("pool", False)       # (models/kan_alexnet.py fuses the MaxPool2d(3, 2) that follows into the layer)
                if args[0].requires_grad:
                    args[0].register_hook(lambda gr, i=i: rec[i].__setitem__("dx", gr.detach()))
            def post(mod, args, out, i=i):
                rec[i]["y"] = out.detach()
                out.register_hook(lambda gr, i=i: rec[i].__setitem__("dy", gr.detach()))
            hooks += [f.register_forward_pre_hook(pre, with_kwargs=True), f.register_forward_hook(post)]
    full_logits, full_dx, full_g = step(0, 128)
    for h in hooks:
        h.remove()
    layer_grads = {i: {n: p.grad.double().clone() for n, p in model.features[i].named_parameters()} for i in rec}

    # (a) the fixture image inside the 128-image launch; logits of chunked launches
    for sfx in ("", "64"):
        ref = d["logits" + sfx].astype(np.float64)
        err = float(np.abs(full_logits[0:1].cpu().numpy() - ref).max() / np.abs(ref).max())
        print(f"[config 5 full] sample 0 logits vs reference fp{sfx or 32}: {err:.2e} (tol {tol['logits']:.1e})")
        assert err <= tol["logits"], (sfx, err)
    sum_g = [torch.zeros_like(v) for v in full_g]
    worst = dict(logits=0.0, dx=0.0)
    for lo, hi in ((0, 1), (1, 32), (32, 64), (64, 128)):
        lg, dx, gs = step(lo, hi)
        worst["logits"] = max(worst["logits"], float((lg - full_logits[lo:hi]).abs().max() / full_logits.abs().max()))
        worst["dx"] = max(worst["dx"], float((dx - full_dx[lo:hi]).abs().max() / full_dx.abs().max()))
        for a, b in zip(sum_g, gs):
            a += b
        if (lo, hi) == (0, 1):                               # (c) the fixture's own step (mean == sum for one image)
            gn = np.array([float(v.norm()) for v in gs])
            rel = np.abs(gn - d["grad_norm"]) / (d["grad_norm"] + 1e-30)
            print(f"[config 5 full] 1-image chunk vs reference gradient norms: {rel.max():.2e} at {names[int(rel.argmax())]} (tol {tol['grad_norm']:.1e})")
            assert rel.max() <= tol["grad_norm"], (names[int(rel.argmax())], rel.max())
    gerr = [float((a - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(sum_g, full_g)]
    k = int(np.argmax(gerr))
    print(f"[config 5 full] model step, full launch vs chunked steps: logits {worst['logits']:.2e} (tol {tol['logits']:.1e}); not asserted (pool / PReLU "
          f"re-routing): dx {worst['dx']:.2e}, gradient max {gerr[k]:.2e} at {names[k]}")
    assert worst["logits"] <= tol["logits"], worst

    # (b) every layer alone on what it saw in the full step
    bad = []
    for i in sorted(rec):
        layer, r = model.features[i], rec[i]

        def rerun(xs, dys):
            layer.zero_grad(set_to_none=True)
            xi = xs.clone().requires_grad_(True)
            y = layer(xi, pool=r["pool"])
            y.backward(dys)
            torch.cuda.synchronize()
            return y.detach(), xi.grad, {n: p.grad.double().clone() for n, p in layer.named_parameters()}
        sums, ey, ex = None, 0.0, 0.0
        for lo in range(0, 128, 32):
            y, dx, gl = rerun(r["x"][lo:lo + 32], r["dy"][lo:lo + 32])
            ey = max(ey, float((y - r["y"][lo:lo + 32]).abs().max() / r["y"].abs().max()))
            if "dx" in r:
                ex = max(ex, float((dx - r["dx"][lo:lo + 32]).abs().max() / r["dx"].abs().max()))
            sums = gl if sums is None else {n: sums[n] + gl[n] for n in gl}
        eg = {n: float((sums[n] - layer_grads[i][n]).abs().max() / (layer_grads[i][n].abs().max() + 1e-30)) for n in sums}
        print(f"[config 5 full] features.{i} {tuple(r['x'].shape)} pool={r['pool']}: full launch vs 4 x 32 images  y {ey:.2e}  dx {ex:.2e}  "
              + "  ".join(f"{n} {v:.2e}" for n, v in eg.items()))
        if r["pool"]:
            # A layer with the MaxPool2d(3, 2) fused behind it is not continuous: the 1e-6 differences between launch configurations flip the choice
            # of a window with two near-equal candidates and re-route its gradient (measured: dx 1e-2, dW 3e-2 max-normalised from a handful of
            # windows).  Values are asserted against the chunks; the gradients through a size-independent property that keeps the launch
            # configuration and therefore every decision: the same 128 images in REVERSED order give the reversed outputs and input gradients
            # and the same weight gradient.
            if ey > 2e-5:
                bad.append((i, "pooled values", ey))
            yp, dxp, gp = rerun(r["x"].flip(0).contiguous(), r["dy"].flip(0).contiguous())
            py = float((yp.flip(0) - r["y"]).abs().max() / r["y"].abs().max())
            px = float((dxp.flip(0) - r["dx"]).abs().max() / r["dx"].abs().max()) if "dx" in r else 0.0
            pg = {n: float((gp[n] - layer_grads[i][n]).abs().max() / (layer_grads[i][n].abs().max() + 1e-30)) for n in gp}
            print(f"[config 5 full] features.{i}: reversed batch  y {py:.2e}  dx {px:.2e}  " + "  ".join(f"{n} {v:.2e}" for n, v in pg.items()))
            if py > 1e-6 or px > 2e-5 or max(pg.values()) > 1e-4:
                bad.append((i, "reversed batch", py, px, pg))
        elif ey > 2e-5 or ex > 2e-5 or max(eg.values()) > 1e-4:
            bad.append((i, ey, ex, eg))
    assert not bad, bad


def _capture(model, layer_type, xin, tt):
    """One train-mode forward + CE + backward; per KAN layer (index in model.features): input x, output y, dL/dy, dL/dx."""
    rec, hooks = {}, []

    def pre(i):
        def f(mod, args):
            rec.setdefault(i, {})["x"] = args[0].detach()
            if args[0].requires_grad:
                args[0].register_hook(lambda g: rec[i].__setitem__("dx", g.detach()))
        return f

    def post(i):
        def f(mod, args, out):
            rec[i]["y"] = out.detach()
            out.register_hook(lambda g: rec[i].__setitem__("dy", g.detach()))
        return f
    for i, f in enumerate(model.features):
        if isinstance(f, layer_type):
            hooks += [f.register_forward_pre_hook(pre(i)), f.register_forward_hook(post(i))]
    model.train()
    logits = model(xin)
    loss = F.cross_entropy(logits, tt)
    loss.backward()
    for h in hooks:
        h.remove()
    return rec, logits.detach(), float(loss.detach())


def test_kan_vgg11_bs256_vs_oracle(gpu_lib):
    """The headline configuration itself (BASELINE.json configs[2]: KAN-VGG11, 256 x 3x32x32, train mode) against the CPU
    oracle with the SAME weights (reference composition: models/kan_vgg.py:178-188 over kan_layers.py:197-247), run in fp64
    (the yardstick) and in fp32 (what the reference arithmetic itself reproduces).  K = K_SPREAD throughout.

    (a) logits / loss, and every layer's input activation: HIP-vs-fp64 <= max(1e-5, K x oracle-fp32-vs-fp64).
    (b) every layer on the HIP model's OWN bs-256 input and output gradient: y, dx and the weight gradients against the fp64
        oracle layer evaluated on exactly those tensors <= max(stated single-layer tolerance, K x what the fp32 oracle layer
        achieves on them).  This is the parity statement proper: it holds to ~1e-6.  The oracle differentiates the PReLU
        branch the HIP layer took (`prelu_gate`): among the 2 M normalised values of a layer a handful lie within rounding
        noise of the kink, and ONE gate taken the other way moves a weight-gradient row by ~1e-2 of its magnitude (a row sums
        4096 signed terms) with both sides correct.
    (c) end-to-end parameter gradients.  Eight InstanceNorm layers over planes down to 2x2 followed by PReLU gates make the
        gradient a DISCONTINUOUS function of the activations (measured on the fp64 model: activation noise 1e-6 moves the last
        layer's weight gradient by 2e-5, noise 1e-5 by 2e-2), so the bound is the exact model's own response: the fp64 oracle
        is re-run with Gaussian noise of the HIP path's measured per-layer activation error (L2) added to each layer input, and
        the HIP gradients must lie within max(1e-3, K x that response) of the fp64 gradients."""
    import copy
    import convkan_amd as K
    from convkan_amd.models import vggkan
    from oracle.kan_oracle import OracleKANConv2d, OracleKANVGG
    torch.manual_seed(0)
    m = vggkan(3, 10, arch="VGG11", kan_conv="KAN", classifier_type="Linear", dropout_linear=0.0)
    m.fuse_pool = False                                  # layer outputs at full size, so that (b) sees each layer's own dL/dy
    o = OracleKANVGG()
    o.classifier[0].p = 0.0
    sd, osd = m.state_dict(), o.state_dict()
    assert [tuple(v.shape) for v in sd.values()] == [tuple(v.shape) for v in osd.values()]
    o.load_state_dict({k: v.clone() for k, v in zip(osd.keys(), sd.values())})      # same registration order, oracle-side names

    def to64(mod):
        mod = copy.deepcopy(mod).double()
        for q in mod.modules():
            if isinstance(getattr(q, "knots", None), torch.Tensor):
                q.knots = q.knots.double()
        return mod
    o64 = to64(o)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(256, 3, 32, 32, generator=g)
    t = torch.randint(0, 10, (256,), generator=g)
    rec32, l32, loss32 = _capture(o, OracleKANConv2d, x, t)
    g32 = [q.grad.double().flatten() for q in o.parameters()]
    rec64, l64, loss64 = _capture(o64, OracleKANConv2d, x.double(), t)
    g64 = [q.grad.flatten().clone() for q in o64.parameters()]
    m = m.cuda()
    rech, lh, lossh = _capture(m, K.KANConvNDLayer, x.cuda(), t.cuda())
    torch.cuda.synchronize()
    dist = lambda a, b: float((a.double().cpu() - b.double().cpu()).norm() / (b.double().norm() + 1e-300))
    dmax = lambda a, b: float((a.double().cpu() - b.double().cpu()).abs().max() / (b.double().abs().max() + 1e-300))
    bad = []

    # ---- (a) forward
    e_hip, e_o32 = dist(lh, l64), dist(l32, l64)
    print(f"[bs256 a] logits HIP-vs-fp64 {e_hip:.2e} (oracle fp32-vs-fp64 {e_o32:.2e});  loss HIP {lossh:.7f} fp32 {loss32:.7f} fp64 {loss64:.7f}")
    assert e_hip <= max(1e-5, K_SPREAD * e_o32)
    assert abs(lossh - loss64) <= max(STATED["loss"], K_SPREAD * abs(loss32 - loss64))
    act_err = {}
    for i in sorted(rech):
        act_err[i] = dist(rech[i]["x"], rec64[i]["x"])
        eo = dist(rec32[i]["x"], rec64[i]["x"])
        print(f"[bs256 a] features.{i} input: HIP-vs-fp64 {act_err[i]:.2e}  oracle fp32-vs-fp64 {eo:.2e}")
        if act_err[i] > max(1e-5, K_SPREAD * eo):
            bad.append(("activation", i, act_err[i], eo))

    # ---- (b) each layer on the HIP model's own input / output gradient
    stated = {"y": 1e-5, "dx": 1e-5, "w_base": 5e-5, "w_spline": 5e-5, "prelu": 5e-5}
    for i in sorted(rech):
        fh = m.features[i]
        got = {"y": rech[i]["y"], "dx": rech[i].get("dx"), "w_base": fh.base_conv[0].weight.grad, "w_spline": fh.spline_conv[0].weight.grad,
               "prelu": fh.prelus[0].weight.grad}
        ref = {}
        for tag, lay, dt in (("f64", copy.deepcopy(o64.features[i]), torch.float64), ("f32", copy.deepcopy(o.features[i]), torch.float32)):
            lay.zero_grad(set_to_none=True)
            xi = rech[i]["x"].to(dt).cpu().requires_grad_(True)
            yo = lay(xi, prelu_gate=(rech[i]["y"] > 0).cpu())
            yo.backward(rech[i]["dy"].to(dt).cpu())
            ref[tag] = {"y": yo.detach(), "dx": xi.grad, **{n: q.grad for n, q in lay.named_parameters()}}
        line = f"[bs256 b] features.{i} on its own inputs:"
        for k in got:
            if got[k] is None:
                continue
            eh = dmax(got[k], ref["f64"][k])
            eo = dmax(ref["f32"][k], ref["f64"][k])               # what the fp32 oracle layer achieves on the same tensors
            line += f"  {k} {eh:.1e} ({eo:.1e})"
            if eh > max(stated[k], K_SPREAD * eo):
                bad.append(("own-input", i, k, eh, eo))
        print(line + "   [HIP-vs-fp64 (oracle fp32-vs-fp64), max-normalised]")

    # ---- (c) end-to-end gradients: the exact (fp64) model's own response to activation noise of the HIP path's measured size is
    # re-measured on every run and K x it is the bound (the response is 2.4e-2 - 3.1e-2 (L2) on the conv weights,
    # HIP sits at 0.8e-2 - 1.1e-2, the fp32 oracle at 1e-4 - 2e-3).
    gen = torch.Generator().manual_seed(7)
    hooks = []
    for i, f in enumerate(o64.features):
        if isinstance(f, OracleKANConv2d) and act_err.get(i, 0.0) > 0.0:
            def noisy(mod, args, i=i):
                a = args[0]
                return (a + act_err[i] * a.pow(2).mean().sqrt() * torch.randn(a.shape, generator=gen, dtype=a.dtype),)
            hooks.append(f.register_forward_pre_hook(noisy))
    o64.zero_grad(set_to_none=True)
    F.cross_entropy(o64(x.double()), t).backward()
    for h in hooks:
        h.remove()
    gn = [q.grad.flatten() for q in o64.parameters()]
    rows = []
    for k, ((n, p), a32, a64) in enumerate(zip(m.named_parameters(), g32, g64)):
        a = p.grad.detach().double().cpu().flatten()
        rows.append((n, dist(a, a64), dist(a32, a64), dist(gn[k], a64), float(a64.norm()),
                     float(torch.dot(a, a64) / (a.norm() * a64.norm() + 1e-300))))
    # the eight PReLU slopes are single numbers, each a sum of millions of signed terms: their individual responses scatter
    # by orders of magnitude from one noise draw to the next, so they share one bound (the largest response among them)
    slope_bound = max(max(eo, er) for n, _, eo, er, _, _ in rows if "prelus" in n)
    for n, eh, eo, er, norm, cos in rows:
        print(f"[bs256 c] {n:34s} |g| {norm:.3e}  HIP-vs-fp64 {eh:.2e}  cosine {cos:.6f}  oracle fp32-vs-fp64 {eo:.2e}"
              f"  fp64 under HIP-sized activation noise {er:.2e}")
        if eh > max(STATED["grad_norm"], K_SPREAD * (slope_bound if "prelus" in n else max(eo, er))):
            bad.append(("gradient", n, eh, eo, er))
    assert not bad, bad


def test_training_step_is_bitwise_deterministic(gpu_lib):
    """Split-K slabs are summed in a fixed order and no float atomics touch the data path (only the PReLU-slope and
    affine-norm gradients use atomics): two identical KAN-VGG11 steps must give bit-identical logits, input gradient and
    conv-weight gradients.  Doubles as a race detector for the LDS-DMA pipelines (a wave reading a tile another wave's
    async copy has not finished shows up here as run-to-run differences)."""
    import torch.nn.functional as F
    from convkan_amd.models import vggkan
    torch.manual_seed(11)
    m = vggkan(3, 10, arch="VGG11", kan_conv="KAN", dropout_linear=0.0).cuda().train()
    x = torch.randn(256, 3, 32, 32, device="cuda")
    t = torch.randint(0, 10, (256,), device="cuda")
    runs = []
    for _ in range(3):
        m.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        logits = m(xi)
        F.cross_entropy(logits, t).backward()
        torch.cuda.synchronize()
        runs.append((logits.detach().clone(), xi.grad.clone(),
                     {n: p.grad.clone() for n, p in m.named_parameters() if p.dim() == 4}))
    for k in (1, 2):
        assert torch.equal(runs[0][0], runs[k][0]), "logits differ between identical runs"
        assert torch.equal(runs[0][1], runs[k][1]), "input gradient differs between identical runs"
        for n, g in runs[0][2].items():
            assert torch.equal(g, runs[k][2][n]), f"{n} gradient differs between identical runs"


def test_gradient_sinks_write_into_the_reducer_buckets(gpu_lib):
    """parallel/dp.py registers bucket slices as gradient sinks: the weight-gradient unpack writes there, autograd adopts the
    view as .grad (no gradient -> bucket copy), values are unchanged, and accumulation over two backward passes still sums."""
    import torch.nn as nn
    import convkan_amd as K
    from convkan_amd.parallel import BucketedGradReducer
    torch.manual_seed(0)
    net = nn.Sequential(K.KANConv2DLayer(4, 128, 3, padding=1), K.KANConv2DLayer(128, 8, 3, padding=1, groups=2)).cuda()
    x = torch.randn(3, 4, 8, 8, device="cuda")
    net(x).square().mean().backward()
    plain = [p.grad.clone() for p in net.parameters()]
    net.zero_grad(set_to_none=True)
    red = BucketedGradReducer(net.parameters())
    try:
        net(x).square().mean().backward()
        red.finish()
        views = {id(p): v for b in red.buckets for p, v in zip(b.params, b.views)}
        in_place = 0
        for (n, p), g in zip(net.named_parameters(), plain):
            err = float((p.grad - g).abs().max()) / (float(g.abs().max()) + 1e-30)
            assert err <= (0.0 if p.dim() == 4 else 1e-5), (n, err)           # PReLU slopes sum with float atomics
            assert p.grad.data_ptr() == views[id(p)].data_ptr(), n            # finish() publishes the bucket views
        net.zero_grad(set_to_none=True)
        hits = []
        hooks = [p.register_post_accumulate_grad_hook(lambda q: hits.append(q.grad.data_ptr() == views[id(q)].data_ptr())) for p in net[0].parameters()
                 if p.dim() == 4]
        with red.no_sync():                                                   # accumulation step: first pass only accumulates
            net(x).square().mean().backward()
        assert hits and all(hits)                                             # single-group conv weights arrived in place
        net(x).square().mean().backward()                                     # second pass accumulates (no sink: .grad is set)
        red.finish()
        for h in hooks:
            h.remove()
        for p, g in zip(net.parameters(), plain):
            assert float((p.grad - 2 * g).abs().max()) <= 1e-5 * float(g.abs().max()) + 1e-12
    finally:
        red.remove()
    from convkan_amd import ops
    assert not ops.GRAD_SINKS


def test_fused_adamw_reads_gradients_from_reducer_buckets(gpu_lib):
    """Reducer (gradient sinks, bucket views published as .grad) + FusedAdamW (gradient address table): the update equals the
    one without a reducer, over three steps."""
    import copy
    import torch.nn as nn
    import convkan_amd as K
    from convkan_amd.parallel import BucketedGradReducer
    torch.manual_seed(0)
    a = nn.Sequential(K.KANConv2DLayer(4, 128, 3, padding=1), K.KANConv2DLayer(128, 8, 3, padding=1)).cuda()
    b = copy.deepcopy(a)
    x = torch.randn(3, 4, 8, 8, device="cuda")
    oa, ob = K.FusedAdamW(a.parameters(), lr=1e-2, weight_decay=1e-2), K.FusedAdamW(b.parameters(), lr=1e-2, weight_decay=1e-2)
    red = BucketedGradReducer(b.parameters())
    try:
        for _ in range(3):
            oa.zero_grad(); ob.zero_grad()
            a(x).square().mean().backward()
            b(x).square().mean().backward()
            red.finish()
            oa.step(); ob.step()
        for (n, p), q in zip(a.named_parameters(), b.parameters()):
            err = float((p - q).abs().max()) / (float(p.abs().max()) + 1e-30)
            assert err <= (1e-6 if p.dim() == 4 else 1e-4), (n, err)          # PReLU slopes: float atomics upstream
    finally:
        red.remove()


def test_fastkan_layer_step_replays_as_hip_graph(gpu_lib):
    """BASELINE.json configs[1] is launch-bound in eager mode (0.24 ms of kernels in a 0.43-0.48 ms step): the whole step -- every launch goes through ctypes
    onto torch's current stream -- must be capturable into ONE HIP graph and replay bit-identically (bench.py reports its replay time under
    other_workloads.fastkan_layer.hip_graph)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    bench._load_torch()
    dev = torch.device("cuda:0")
    model = bench.build_model(dev, "fastkan_layer")
    x = torch.randn(256, 3, 32, 32, device=dev)
    res = bench.hip_graph_replay(model, x, None, iters=5)
    assert res.get("matches_eager_bitwise") is True, res


@pytest.mark.parametrize("kind", ["KAN", "FastKAN", "ChebyKAN"])
def test_graphed_training_steps_equal_eager_steps_bitwise(kind, gpu_lib):
    """train.GraphedStep: forward + loss + backward recorded once into a HIP graph, FusedAdamW outside it.  Six training steps on changing batches must
    leave exactly the weights and losses of six eager `train_step`s from the same initialisation -- in particular the recorded weight packs must read
    the weights the optimizer wrote between replays (ops.always_pack), and the recorded backward must assign, not accumulate, the gradients."""
    import copy
    import convkan_amd as K
    from convkan_amd.models import vggkan
    torch.manual_seed(7)
    base = vggkan(3, 10, arch="VGG11", kan_conv=kind, dropout_linear=0.0).cuda().train()
    g = torch.Generator(device="cuda").manual_seed(3)
    batches = [(torch.randn(32, 3, 32, 32, device="cuda", generator=g), torch.randint(0, 10, (32,), device="cuda", generator=g)) for _ in range(6)]
    out = {}
    for mode in ("eager", "graph"):
        model = copy.deepcopy(base)
        opt = K.FusedAdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
        losses = []
        if mode == "eager":
            for d, t in batches:
                losses.append(float(K.train_step(model, d, t, opt)))
        else:
            step = K.GraphedStep(model, batches[0][0], batches[0][1])
            for d, t in batches:
                losses.append(float(step(d, t)))
                opt.step()
        torch.cuda.synchronize()
        out[mode] = (losses, [p.detach().clone() for p in model.parameters()])
    assert out["eager"][0] == out["graph"][0], (out["eager"][0], out["graph"][0])
    for a, b in zip(out["eager"][1], out["graph"][1]):
        assert torch.equal(a, b)
    assert out["eager"][0][-1] != out["eager"][0][0]          # (the weights did move)
