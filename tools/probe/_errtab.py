import sys, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from conftest import golden_cases, load_golden
from helpers import build_layer, relerr
import convkan_amd
for name in golden_cases():
    d = load_golden(name); c = d['cfg']
    layer = build_layer(c)
    layer.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in d.items() if k.startswith('sd.')})
    layer = layer.cuda().train()
    x = torch.from_numpy(d['x']).cuda().requires_grad_(True)
    y = layer(x); y.backward(torch.from_numpy(d['g']).cuda())
    e = {'y': relerr(y, torch.from_numpy(d['y'])), 'dx': relerr(x.grad, torch.from_numpy(d['dx']))}
    for n, p in layer.named_parameters():
        if 'grad.'+n in d: e[n.replace('.weight','').replace('_conv','')] = relerr(p.grad, torch.from_numpy(d['grad.'+n]))
    print(f"{name:22s}", ' '.join(f"{k}={v:.1e}" for k, v in e.items()))
