import sys, torch, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from conftest import load_golden
from helpers import build_layer
d = load_golden('bspline_vgg_l6'); c = d['cfg']
layer = build_layer(c); layer.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in d.items() if k.startswith('sd.')}); layer = layer.cuda()
x = torch.from_numpy(d['x']).cuda().requires_grad_(True)
y = layer(x); y.backward(torch.from_numpy(d['g']).cuda())
e = (x.grad.cpu() - torch.from_numpy(d['dx'])).abs(); ref = torch.from_numpy(d['dx'])
print('max|dx|', float(ref.abs().max()), 'max err', float(e.max()), 'median err', float(e.median()), '99pct', float(e.flatten().kthvalue(int(e.numel()*0.99)).values))
idx = torch.topk(e.flatten(), 8).indices
for i in idx.tolist():
    print('x=%.6f err=%.2e dx_ref=%.4e' % (float(torch.from_numpy(d['x']).flatten()[i]), float(e.flatten()[i]), float(ref.flatten()[i])))
print([tuple(int(v) for v in np.unravel_index(i, e.shape)) for i in torch.topk(e.flatten(), 24).indices.tolist()])
ec = e.amax(dim=(0, 2, 3)); print('per-channel max err', [f"{v:.1e}" for v in ec.tolist()])
eb = e.amax(dim=(1, 2, 3)); print('per-image max err', [f"{v:.1e}" for v in eb.tolist()])
