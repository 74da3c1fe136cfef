"""Diagnostic: per-layer activation error of the HIP KAN-VGG11 vs an fp64 CPU oracle, beside the fp32 CPU oracle's."""
import sys, os, numpy as np, torch, torch.nn as nn, torch.nn.functional as F
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import convkan_amd as K
from convkan_amd.models import vggkan
from oracle import kan_oracle as O
from test_gpu_models import model_fill
d = np.load('tests/golden/model_kan_vgg11.npz')
x = torch.from_numpy(d['x'])
m = vggkan(3, 10, arch='VGG11', kan_conv='KAN', classifier_type='Linear'); model_fill(m); m.eval()
def oracle_acts(dt):
    h = x.to(dt); acts = []
    for f in m.features:
        if isinstance(f, nn.MaxPool2d): h = F.max_pool2d(h, 2, 2)
        else:
            h = O.kan_conv2d(h, [f.base_conv[0].weight.detach().to(dt)], [f.spline_conv[0].weight.detach().to(dt)], [f.prelus[0].weight.detach().to(dt)],
                             knots=f.grid.to(dt), spline_order=3, act=F.silu, padding=1)
        acts.append(h)
    return acts
a64 = oracle_acts(torch.float64); a32 = oracle_acts(torch.float32)
mg = m.cuda(); h = x.cuda(); ag = []
with torch.no_grad():
    for f in mg.features:
        h = f(h); ag.append(h)
rel = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max())
for i, f in enumerate(m.features):
    print(i, type(f).__name__, tuple(a64[i].shape), f"cpu32 {rel(a32[i], a64[i]):.1e}  hip {rel(ag[i], a64[i]):.1e}  min plane std {float(a64[i].flatten(2).std(-1).min()) if a64[i].shape[-1]>1 else 0:.2e}")
# ---- focus on feature 10 (first 2x2 layer): same fp64-derived input to all paths
f = m.features[10]; fin64 = a64[9]
def lay(dt, inp):
    pre = []
    y = O.kan_conv2d(inp.to(dt), [f.base_conv[0].weight.detach().cpu().to(dt)], [f.spline_conv[0].weight.detach().cpu().to(dt)], [f.prelus[0].weight.detach().cpu().to(dt)],
                     knots=f.grid.to(dt), spline_order=3, act=F.silu, padding=1, pre_norm_out=pre)
    return pre[0], y
z64, y64 = lay(torch.float64, fin64)
z32, y32 = lay(torch.float32, fin64)
with torch.no_grad():
    fg = mg.features[10]
    xin = fin64.float().cuda()
    zg = K.ops.kan_conv(fg.conv_spec(), xin, None, [fg.base_conv[0].weight], [fg.spline_conv[0].weight]).cpu()
    yg = fg(xin).cpu()
print("same-input layer10: z cpu32 %.1e hip %.1e | y cpu32 %.1e hip %.1e" % (rel(z32, z64), rel(zg, z64), rel(y32, y64), rel(yg, y64)))
pstd = z64.flatten(2).std(-1, unbiased=False); print("z plane std: min %.3e median %.3e max|z| %.3e" % (float(pstd.min()), float(pstd.median()), float(z64.abs().max())))
ey = (yg.double() - y64).abs().flatten(2).max(-1).values; idx = ey.argmax(); b, o = int(idx // ey.shape[1]), int(idx % ey.shape[1])
print("worst hip plane", b, o, "err", float(ey[b, o]), "plane std", float(pstd[b, o]), "z64", z64[b, o].flatten().tolist(), "zg", zg[b, o].flatten().tolist(), "z32", z32[b, o].flatten().tolist())
ec = (y32.double() - y64).abs().flatten(2).max(-1).values; print("cpu32 err at that plane", float(ec[b, o]), "cpu32 worst", float(ec.max()), "at std", float(pstd.flatten()[ec.argmax()]))
rms = lambda a, b: float(((a.double().cpu() - b) ** 2).mean().sqrt() / (b ** 2).mean().sqrt())
for i in range(len(a64)):
    print(i, "rms cpu32 %.1e hip %.1e | max cpu32 %.1e hip %.1e" % (rms(a32[i], a64[i]), rms(ag[i], a64[i]), rel(a32[i], a64[i]), rel(ag[i], a64[i])))
_, yp_h = lay(torch.float64, ag[9].double().cpu()); _, yp_c = lay(torch.float64, a32[9].double())
print("fp64 layer10 on hip's input: %.1e ; on cpu32's input: %.1e" % (rel(yp_h, y64), rel(yp_c, y64)))
e9 = (ag[9].double().cpu() - a64[9]).abs(); print("hip layer9 err: max", float(e9.max()), "at", np.unravel_index(int(e9.argmax()), e9.shape), "99.9pct", float(e9.flatten().kthvalue(int(e9.numel()*0.999)).values))
e9c = (a32[9].double() - a64[9]).abs(); print("cpu layer9 err: max", float(e9c.max()), "99.9pct", float(e9c.flatten().kthvalue(int(e9c.numel()*0.999)).values))
for i in (8, 9):
    e = (ag[i].double().cpu() - a64[i]); ec = (a32[i].double() - a64[i])
    print(i, "hip err sample", [f"{v:+.1e}" for v in e[0, :3].flatten()[:12].tolist()], "vals", [f"{v:+.2f}" for v in a64[i][0, :3].flatten()[:12].tolist()])
    print(i, "cpu err sample", [f"{v:+.1e}" for v in ec[0, :3].flatten()[:12].tolist()])
    print(i, "hip |err| quantiles", [float(e.abs().flatten().kthvalue(max(1, int(e.numel() * q))).values) for q in (0.1, 0.5, 0.9, 0.99)])
    neg = a64[i] < 0
    print(i, "hip mean|err| on y<0: %.2e  on y>0: %.2e ; relative err on y>0 median %.2e" % (float(e[neg].abs().mean()) if neg.any() else 0, float(e[~neg].abs().mean()), float((e[~neg] / a64[i][~neg]).abs().median())))
