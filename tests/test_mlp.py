"""MLP KANLayer / KAN head: oracle vs the reference's golden vectors and host-side API."""
import pytest
import torch
import torch.nn as nn

from conftest import golden_cases, load_golden
from helpers import ACT_FN, ACTS, relerr
from oracle import kan_oracle as O


@pytest.mark.parametrize("name", golden_cases("mlp"))
def test_oracle_kan_linear_matches_golden(name):
    d = load_golden(name)
    c = d["cfg"]
    t = lambda k: torch.from_numpy(d[k])
    x = t("x").requires_grad_(True)
    ps = {k: t("sd." + k).requires_grad_(True) for k in ("base_weight", "spline_weight", "layer_norm.weight", "layer_norm.bias", "prelu.weight")}
    pre = []
    y = O.kan_linear(x, ps["base_weight"], ps["spline_weight"], ps["layer_norm.weight"], ps["layer_norm.bias"], ps["prelu.weight"],
                     grid_size=c["G"], spline_order=c["S"], grid_range=c["rng"], act=ACT_FN[c["act"]], pre=pre)
    y.backward(t("g"))
    assert relerr(y, t("y")) < 2e-6 and relerr(pre[0], t("z")) < 2e-6 and relerr(x.grad, t("dx")) < 2e-6
    for k, p in ps.items():
        assert relerr(p.grad, t("grad." + k)) < 5e-6, k


def test_kanlayer_api_mirrors_reference():
    import convkan_amd as K
    d = load_golden("mlp_tiny")
    c = d["cfg"]
    layer = K.KANLayer(c["I"], c["O"], grid_size=c["G"], spline_order=c["S"], base_activation=ACTS[c["act"]], grid_range=c["rng"])
    sd = {k[3:]: torch.from_numpy(v) for k, v in d.items() if k.startswith("sd.")}
    layer.load_state_dict(sd, strict=True)                      # same keys / shapes as kan_layers.py:22-27
    assert [n for n, _ in layer.named_parameters()] == ["base_weight", "spline_weight", "layer_norm.weight", "layer_norm.bias", "prelu.weight"]
    assert tuple(layer.grid.shape) == (c["I"], c["G"] + 2 * c["S"] + 1)
    assert torch.equal(layer.grid[0], O.bspline_knots(c["G"], c["S"], c["rng"]))
    with pytest.raises(Exception):                              # no CPU fallback
        layer(torch.zeros(2, c["I"]))


def test_kan_mlp_and_vgg_heads_structure():
    import convkan_amd as K
    from convkan_amd.models import vggkan
    m = K.mlp_kan([8, 16, 4], dropout=0.1)
    kinds = [type(l).__name__ for l in m.layers]
    assert kinds == ["Dropout", "KANLayer", "Dropout", "KANLayer"]          # models/kans.py:311-321
    assert K.MLP_KAN_FACTORY["KAN"] is K.mlp_kan
    with pytest.raises(NotImplementedError):
        K.mlp_kan([4, 4], l1_decay=0.1)
    for head, first in (("KAN", "1.layers.0.base_weight"), ("HiddenKAN", "0.layers.0.base_weight"), ("VGGKAN", "0.weight")):
        v = vggkan(3, 10, arch="VGG11_kansmall" if False else "VGG11", kan_conv="KAN", classifier_type=head)
        names = [n for n, _ in v.classifier.named_parameters()]
        assert names[0] == first, (head, names)
        kl = [mod for mod in v.classifier.modules() if isinstance(mod, K.KANLayer)]
        assert len(kl) == 1 and isinstance(kl[0].base_activation, nn.SiLU) and kl[0].spline_order == 3
    with pytest.raises(NotImplementedError):
        vggkan(3, 10, arch="VGG11", kan_conv="KAN", classifier_type="KAN", kan_classifier="FastKAN")
