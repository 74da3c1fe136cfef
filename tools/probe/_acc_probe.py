"""Diagnostic (not a test): accuracy of the HIP layer vs the oracle's fp32 CPU path, both measured against an fp64 oracle run."""
import sys, torch, torch.nn as nn, torch.nn.functional as F
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import convkan_amd as K
from oracle import kan_oracle as O
torch.manual_seed(0)
def rel(a, b): return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max())
for (C, Oc, H, B) in [(64, 128, 16, 4), (256, 256, 8, 8), (512, 512, 4, 16), (512, 512, 2, 64)]:
    layer = K.KANConv2DLayer(C, Oc, 3, padding=1, base_activation=nn.SiLU)
    x = torch.randn(B, C, H, H); g = torch.randn(B, Oc, H, H)
    def run(dt):
        xx = x.detach().clone().to(dt).requires_grad_(True)
        pre = []
        y = O.kan_conv2d(xx, [layer.base_conv[0].weight.detach().to(dt).requires_grad_(True)], [layer.spline_conv[0].weight.detach().to(dt).requires_grad_(True)],
                         [layer.prelus[0].weight.detach().to(dt)], knots=layer.grid.to(dt), spline_order=3, act=F.silu, padding=1, pre_norm_out=pre)
        y.backward(g.to(dt))
        return pre[0].detach(), y.detach(), xx.grad
    z64, y64, dx64 = run(torch.float64)
    z32, y32, dx32 = run(torch.float32)
    lg = layer.cuda()
    xg = x.detach().clone().cuda().requires_grad_(True)
    zg = K.ops.kan_conv(lg.conv_spec(), xg.detach(), None, [lg.base_conv[0].weight.detach()], [lg.spline_conv[0].weight.detach()])
    yg = lg(xg); yg.backward(g.cuda())
    print(f"C{C} O{Oc} H{H} B{B}: z cpu32 {rel(z32,z64):.1e} hip {rel(zg,z64):.1e} | y cpu32 {rel(y32,y64):.1e} hip {rel(yg,y64):.1e} | dx cpu32 {rel(dx32,dx64):.1e} hip {rel(xg.grad,dx64):.1e}")
