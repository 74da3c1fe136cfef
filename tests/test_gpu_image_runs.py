"""Batches whose activations pass the 2 GiB launch limit (32-bit buffer offsets) are cut into runs of whole images
(ops._image_runs): the cut is checked against the oracle with a lowered limit, and once for real above 2 GiB."""
import pytest
import torch
import torch.nn as nn

import convkan_amd as K
from convkan_amd import ops
from convkan_amd._lib import KanConvError
from helpers import TOL_DW, TOL_DX, TOL_Y, oracle_forward, relerr

pytestmark = pytest.mark.gpu


def _fwd_bwd(layer, x_cpu, go_cpu):
    layer.zero_grad(set_to_none=True)
    x = x_cpu.clone().cuda().requires_grad_(True)
    y = layer(x)
    y.backward(go_cpu.cuda())
    torch.cuda.synchronize()
    return y.detach(), x.grad.detach(), {n: p.grad.detach().clone() for n, p in layer.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("make,cfg", [
    (lambda: K.KANConv2DLayer(6, 10, 3, padding=1, base_activation=nn.SiLU), dict(kind="bspline", C=6, O=10, k=3, s=1, p=1, d=1, groups=1, act="silu")),
    (lambda: K.FastKANConv2DLayer(6, 10, 3, padding=1, groups=2), dict(kind="rbf", C=6, O=10, k=3, s=1, p=1, d=1, groups=2)),
    (lambda: K.ChebyKANConv2DLayer(6, 10, 3, padding=1, degree=4), dict(kind="cheby", C=6, O=10, k=3, s=1, p=1, d=1, groups=1, degree=4)),
], ids=["kan", "fastkan_g2", "cheby"])
def test_cut_batches_match_the_oracle_and_the_single_launch(make, cfg, gpu_lib, monkeypatch):
    torch.manual_seed(5)
    layer = make()
    x = torch.randn(7, 6, 12, 12)
    xo = x.clone().requires_grad_(True)
    yo = oracle_forward(cfg, layer, xo)
    go = torch.randn(yo.shape, generator=torch.Generator().manual_seed(6))
    yo.backward(go)
    ref = {n: p.grad.clone() for n, p in layer.named_parameters() if p.grad is not None}
    dev = layer.cuda()
    y1, dx1, dw1 = _fwd_bwd(dev, x, go)
    per_image = 4 * 10 * 12 * 12
    monkeypatch.setattr(ops, "MAX_TENSOR_BYTES", 3 * per_image)                 # 7 images -> runs of 3, 3, 1
    assert ops._image_runs(dev.conv_spec(), x, 10) == 3
    y2, dx2, dw2 = _fwd_bwd(dev, x, go)
    assert torch.equal(y1, y2) and torch.equal(dx1, dx2)                       # per-image work is identical launch by launch
    assert relerr(y2, yo) <= TOL_Y and relerr(dx2, xo.grad) <= TOL_DX
    for n, g in dw2.items():
        assert relerr(g, ref[n]) <= (TOL_DW if g.dim() == 4 else 2e-5), n
        assert relerr(g, dw1[n]) <= 1e-5, n
    monkeypatch.setattr(ops, "MAX_TENSOR_BYTES", per_image - 1)
    with pytest.raises(KanConvError):
        dev(x.cuda())


def test_batch_above_2gib_runs_in_two_launches(gpu_lib):
    """4 x 8 x 4096 x 4096 fp32 = 2 GiB exactly: one byte too many for one launch.  Compared with the same images sent one by one."""
    torch.manual_seed(7)
    layer = K.KANConv2DLayer(8, 8, 3, padding=1, base_activation=nn.SiLU).cuda()
    x = torch.randn(4, 8, 4096, 4096, device="cuda")
    assert x.numel() * 4 > ops.MAX_TENSOR_BYTES and ops._image_runs(layer.conv_spec(), x, 8) == 2
    xr = x.clone().requires_grad_(True)
    y = layer(xr)
    w = torch.randn(1, 8, 1, 1, device="cuda")
    (y * w).sum().backward()
    dw = {n: p.grad.clone() for n, p in layer.named_parameters()}
    layer.zero_grad(set_to_none=True)
    for i in range(4):
        xi = x[i:i + 1].clone().requires_grad_(True)
        yi = layer(xi)
        (yi * w).sum().backward()
        assert torch.equal(yi, y[i:i + 1]) and torch.equal(xi.grad, xr.grad[i:i + 1])
    for n, p in layer.named_parameters():
        assert relerr(dw[n], p.grad) <= 1e-5, n
