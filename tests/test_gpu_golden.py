"""GPU parity against the golden vectors frozen from the reference (tests/golden/*.npz).

Each case loads the reference's state_dict into this repo's layer (strict), runs forward + backward through the
C-ABI HIP kernels on cuda:0 and compares y, dx and every parameter gradient, max-normalised per tensor."""
import pytest
import torch

from conftest import golden_cases, load_golden
from helpers import TOL_DW, TOL_DX, TOL_Y, build_layer, relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", golden_cases())
def test_layer_matches_reference(name, gpu_lib):
    d = load_golden(name)
    c = d["cfg"]
    layer = build_layer(c)
    sd = {k[3:]: torch.from_numpy(v) for k, v in d.items() if k.startswith("sd.")}
    layer.load_state_dict(sd, strict=True)
    layer = layer.cuda().train()
    x = torch.from_numpy(d["x"]).cuda().requires_grad_(True)
    y = layer(x)
    assert tuple(y.shape) == d["y"].shape
    y.backward(torch.from_numpy(d["g"]).cuda())
    torch.cuda.synchronize()
    noise = d["noise"]          # the reference's own fp32-vs-fp64 noise per tensor (make_golden.py)

    def tol(key, base):
        return max(base, 4.0 * noise.get(key, 0.0))
    errs = {"y": (relerr(y, torch.from_numpy(d["y"])), tol("y", TOL_Y)), "dx": (relerr(x.grad, torch.from_numpy(d["dx"])), tol("dx", TOL_DX))}
    for n, p in layer.named_parameters():
        key = "grad." + n
        if key in d:
            assert p.grad is not None, n
            errs[n] = (relerr(p.grad, torch.from_numpy(d[key])), tol(key, TOL_DW if p.dim() == 4 else 2e-5))
        else:
            assert p.grad is None or not p.requires_grad or float(p.grad.abs().max()) == 0.0, n
    bad = {k: v for k, v in errs.items() if not v[0] <= v[1]}
    assert not bad, f"{name}: {bad}  (all: {errs})"
