import sys, numpy as np, torch, torch.nn as nn, torch.nn.functional as F
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import convkan_amd as K
from convkan_amd.models import alexnet_kan
from oracle import kan_oracle as O
from test_gpu_models import model_fill
d = np.load('tests/golden/model_cheby_alexnet.npz')
x = torch.from_numpy(d['x'])
m = alexnet_kan(num_classes=10, kan_conv='ChebyKAN', degree=4); model_fill(m); m.eval()
def oracle_acts(dt):
    h = x.to(dt); acts = []
    for f in m.features:
        if isinstance(f, nn.MaxPool2d): h = F.max_pool2d(h, 3, 2)
        else:
            nrm = [lambda z, f=f: F.instance_norm(z, weight=f.layer_norm[0].weight.detach().to(dt), bias=f.layer_norm[0].bias.detach().to(dt))]
            h = O.chebykan_conv2d(h, [f.poly_conv[0].weight.detach().to(dt)], degree=4, stride=f.stride, padding=f.padding, norm=nrm)
        acts.append(h)
    return acts
a64 = oracle_acts(torch.float64); a32 = oracle_acts(torch.float32)
mg = m.cuda(); h = x.cuda(); ag = []
with torch.no_grad():
    for f in mg.features:
        h = f(h); ag.append(h)
rel = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max())
for i, f in enumerate(m.features):
    print(i, type(f).__name__, tuple(a64[i].shape), f"cpu32 {rel(a32[i], a64[i]):.1e}  hip {rel(ag[i], a64[i]):.1e}  max|a| {float(a64[i].abs().max()):.2e}")
lg = mg(x.cuda()); print('logits hip vs golden', rel(lg.detach(), torch.from_numpy(d['logits']).double()))
# ---- backward: oracle CPU fp32 autograd vs HIP, same parameters
t = torch.from_numpy(d['t'])
mc = alexnet_kan(num_classes=10, kan_conv='ChebyKAN', degree=4); model_fill(mc); mc.eval()
h = x.clone()
for f in mc.features:
    if isinstance(f, nn.MaxPool2d): h = F.max_pool2d(h, 3, 2)
    else:
        nrm = [lambda z, f=f: F.instance_norm(z, weight=f.layer_norm[0].weight, bias=f.layer_norm[0].bias)]
        h = O.chebykan_conv2d(h, [f.poly_conv[0].weight], degree=4, stride=f.stride, padding=f.padding, norm=nrm)
lo = mc.classifier(torch.flatten(mc.avgpool(h), 1)); F.cross_entropy(lo, t).backward()
mg.zero_grad(set_to_none=True)
lg = mg(x.cuda()); F.cross_entropy(lg, t.cuda()).backward()
gold_gn = d['grad_norm']
for i, ((n, pc), (_, pg)) in enumerate(zip(mc.named_parameters(), mg.named_parameters())):
    print(f"{n:36s} |g| oracle {float(pc.grad.norm()):.4e} golden {gold_gn[i]:.4e} hip {float(pg.grad.norm()):.4e}  maxnorm err hip-vs-oracle {rel(pg.grad, pc.grad.double()):.1e}")
# ---- isolate the feature stack: same random upstream gradient through oracle (CPU) and HIP features
gen = torch.Generator().manual_seed(5); Gup = torch.randn(1, 256, 6, 6, generator=gen)
mc.zero_grad(set_to_none=True); mg.zero_grad(set_to_none=True)
h = x.clone(); acts_c = []
for f in mc.features:
    if isinstance(f, nn.MaxPool2d): h = F.max_pool2d(h, 3, 2)
    else:
        nrm = [lambda z, f=f: F.instance_norm(z, weight=f.layer_norm[0].weight, bias=f.layer_norm[0].bias)]
        h = O.chebykan_conv2d(h, [f.poly_conv[0].weight], degree=4, stride=f.stride, padding=f.padding, norm=nrm)
    h.retain_grad(); acts_c.append(h)
h.backward(Gup)
hg = x.cuda(); acts_g = []
for f in mg.features:
    hg = f(hg); hg.retain_grad(); acts_g.append(hg)
hg.backward(Gup.cuda())
for i in range(len(acts_c)):
    print(i, type(mc.features[i]).__name__, 'd(act) err', f"{rel(acts_g[i].grad, acts_c[i].grad.double()):.1e}")
for (n, pc), (_, pg) in zip(mc.features.named_parameters(), mg.features.named_parameters()):
    print(f"{n:28s} err {rel(pg.grad, pc.grad.double()):.1e}")
# ---- ReLU gate analysis with the real features
with torch.no_grad():
    fc = mc.classifier; fg = mg.classifier
    ac = torch.flatten(mc.avgpool(a32[7]), 1); agp = torch.flatten(mg.avgpool(ag[7]), 1)
    p1c = fc.fc1(ac); p1g = fg.fc1(agp).cpu()
    print('fc1 pre-act: err', rel(p1g, p1c.double()), 'min|v| cpu', float(p1c.abs().min()), 'flips', int(((p1g > 0) != (p1c > 0)).sum()), 'frac<1e-4', float((p1c.abs() < 1e-4).float().mean()), 'frac>0', float((p1c > 0).float().mean()))
    h1c = torch.relu(p1c); p2c = fc.fc2(h1c); p2g = fg.fc2(torch.relu(p1g.cuda())).cpu()
    print('fc2 pre-act: err', rel(p2g, p2c.double()), 'min|v| cpu', float(p2c.abs().min()), 'flips', int(((p2g > 0) != (p2c > 0)).sum()), 'frac>0', float((p2c > 0).float().mean()))
# gradient wrt features from both classifiers, given their own features
fcpu = a32[7].clone().requires_grad_(True); F.cross_entropy(mc.classifier(torch.flatten(mc.avgpool(fcpu), 1)), t).backward()
fgpu = ag[7].clone().requires_grad_(True); F.cross_entropy(mg.classifier(torch.flatten(mg.avgpool(fgpu), 1)), t.cuda()).backward()
print('d loss / d features: hip-vs-cpu', rel(fgpu.grad, fcpu.grad.double()), ' norms', float(fgpu.grad.norm()), float(fcpu.grad.norm()))
# feed the CPU features to the GPU classifier
fg2 = a32[7].clone().cuda().requires_grad_(True); F.cross_entropy(mg.classifier(torch.flatten(mg.avgpool(fg2), 1)), t.cuda()).backward()
print('same (cpu) features, gpu classifier vs cpu classifier:', rel(fg2.grad, fcpu.grad.double()))
