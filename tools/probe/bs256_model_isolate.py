#!/usr/bin/env python3
"""Where does the bs-256 KAN-VGG11 weight-gradient error come from?  Captures every layer's input and output gradient in the HIP
model and in the fp64 oracle model (same weights), compares them, then re-runs each oracle layer in fp64 on the HIP model's OWN
layer inputs / output gradients: isolates 'the layer is inaccurate on this data' from 'its inputs already differ'."""
import copy, os, sys
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from convkan_amd.models import vggkan
from oracle.kan_oracle import OracleKANVGG, OracleKANConv2d

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
m = vggkan(3, 10, arch="VGG11", kan_conv="KAN", classifier_type="Linear", dropout_linear=0.0)
if os.environ.get("NO_FUSE_POOL"):
    m.fuse_pool = False
o = OracleKANVGG(); o.classifier[0].p = 0.0
o.load_state_dict({k: v.clone() for k, v in zip(o.state_dict().keys(), m.state_dict().values())})
o64 = copy.deepcopy(o).double()
for mod in o64.modules():
    if isinstance(getattr(mod, "knots", None), torch.Tensor): mod.knots = mod.knots.double()
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 3, 32, 32, generator=g); t = torch.randint(0, 10, (B,), generator=g)

def capture(model, layer_types, xin, tt):
    rec = {}
    hooks = []
    for i, f in enumerate(model.features):
        if isinstance(f, layer_types):
            hooks.append(f.register_forward_hook(lambda mod, a, out, i=i: (rec.setdefault(i, {}).update(x=a[0].detach()), out.register_hook(lambda gr, i=i: rec[i].update(dy=gr.detach())))[0] and None))
    model.train()
    loss = F.cross_entropy(model(xin), tt); loss.backward()
    for h in hooks: h.remove()
    return rec
import convkan_amd as K
rec64 = capture(o64, (OracleKANConv2d,), x.double(), t)
rec32 = capture(o, (OracleKANConv2d,), x, t)
g64 = [p.grad.clone() for p in o64.parameters()]
for i in sorted(rec32):
    print(f"oracle fp32 vs fp64  features.{i}: x l2 {float((rec32[i]['x'].double() - rec64[i]['x']).norm() / rec64[i]['x'].norm()):.2e} "
          f"dy l2 {float((rec32[i]['dy'].double() - rec64[i]['dy']).norm() / rec64[i]['dy'].norm()):.2e}", flush=True)
# response of the exact (fp64) model to elementwise relative noise on ONE layer's input: last layer
for eps in (1e-7, 1e-6, 1e-5):
    lay = copy.deepcopy(o64.features[11]); lay.zero_grad()
    gen = torch.Generator().manual_seed(5)
    xi = rec64[11]["x"] * (1.0 + eps * torch.randn(rec64[11]["x"].shape, generator=gen, dtype=torch.float64))
    lay(xi).backward(rec64[11]["dy"])
    ref = o64.features[11].w_spline.grad
    print(f"fp64 last layer, input noise {eps:.0e} (relative, elementwise) -> dW_spline l2 change {float((lay.w_spline.grad - ref).norm() / ref.norm()):.2e}", flush=True)
m = m.cuda()
rech = capture(m, (K.KANConvNDLayer,), x.cuda(), t.cuda())
torch.cuda.synchronize()
l2 = lambda a, b: float((a.double().cpu() - b.double().cpu()).norm() / (b.double().norm() + 1e-300))
mxe = lambda a, b: float((a.double().cpu() - b.double().cpu()).abs().max() / (b.double().abs().max() + 1e-300))
for i in sorted(rech):
    fh, fo = m.features[i], o64.features[i]
    xh, dyh = rech[i]["x"], rech[i]["dy"]
    pooled = dyh.shape != rec64[i]["dy"].shape
    line = f"features.{i}: x l2 {l2(xh, rec64[i]['x']):.2e} max {mxe(xh, rec64[i]['x']):.2e}"
    if not pooled: line += f" | dy l2 {l2(dyh, rec64[i]['dy']):.2e} max {mxe(dyh, rec64[i]['dy']):.2e}"
    # oracle layer (fp64) on the HIP model's own input / output gradient
    lay = copy.deepcopy(fo)
    for p in lay.parameters(): p.grad = None
    xi = xh.double().cpu().requires_grad_(True)
    yo = lay(xi)
    dy_full = dyh.double().cpu()
    if pooled:                                            # HIP layer ran with the fused 2x2 max-pool: route through the same pool
        yo = F.max_pool2d(yo, 2, 2)
    yo.backward(dy_full)
    wh = {"w_base": fh.base_conv[0].weight.grad, "w_spline": fh.spline_conv[0].weight.grad, "prelu": fh.prelus[0].weight.grad}
    for n, p in lay.named_parameters():
        line += f" | {n} own-input l2 {l2(wh[n], p.grad):.2e}"
    print(line, flush=True)
