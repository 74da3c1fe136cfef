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


@pytest.mark.parametrize("name", golden_cases("mlp"))
def test_mlp_kanlayer_matches_reference(name, gpu_lib):
    """MLP KANLayer (kan_layers.py:48-114) on the 1x1 conv stage vs the reference's frozen forward/backward."""
    import convkan_amd as K
    from helpers import ACTS
    d = load_golden(name)
    c = d["cfg"]
    layer = K.KANLayer(c["I"], c["O"], grid_size=c["G"], spline_order=c["S"], base_activation=ACTS[c["act"]], grid_range=c["rng"])
    layer.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in d.items() if k.startswith("sd.")}, strict=True)
    layer = layer.cuda().train()
    x = torch.from_numpy(d["x"]).cuda().requires_grad_(True)
    y = layer(x)
    y.backward(torch.from_numpy(d["g"]).cuda())
    torch.cuda.synchronize()
    noise = d["noise"]
    tol = lambda key, base: max(base, 4.0 * noise.get(key, 0.0))
    errs = {"y": (relerr(y, torch.from_numpy(d["y"])), tol("y", TOL_Y)), "dx": (relerr(x.grad, torch.from_numpy(d["dx"])), tol("dx", TOL_DX))}
    for n, p in layer.named_parameters():
        errs[n] = (relerr(p.grad, torch.from_numpy(d["grad." + n])), tol("grad." + n, TOL_DW))
    bad = {k: v for k, v in errs.items() if not v[0] <= v[1]}
    assert not bad, f"{name}: {bad}  (all: {errs})"


def test_vgg_kan_head_trains(gpu_lib):
    """KAN-VGG11 with the KAN MLP head (kan_vgg.py:134-138) vs the oracle head on the same features."""
    from convkan_amd.models import vggkan
    from oracle import kan_oracle as O
    import torch.nn.functional as F
    torch.manual_seed(3)
    m = vggkan(3, 10, arch="VGG11", kan_conv="KAN", classifier_type="KAN", dropout_linear=0.0).cuda().train()
    x = torch.randn(4, 3, 32, 32, device="cuda")
    feats = torch.flatten(m.avgpool(m.forward_features(x)), 1)
    logits = m(x)
    assert logits.shape == (4, 10) and torch.isfinite(logits).all()
    F.cross_entropy(logits, torch.tensor([1, 2, 3, 4], device="cuda")).backward()
    kl = m.classifier[1].layers[0]
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    f = feats.detach().cpu()
    want = O.kan_linear(f, kl.base_weight.detach().cpu(), kl.spline_weight.detach().cpu(), kl.layer_norm.weight.detach().cpu(),
                        kl.layer_norm.bias.detach().cpu(), kl.prelu.weight.detach().cpu(), grid_size=kl.grid_size,
                        spline_order=kl.spline_order, grid_range=kl.grid_range, act=F.silu)
    assert relerr(logits, want) < 1e-4
