"""CPU: the oracle (oracle/kan_oracle.py) against every golden vector frozen from the reference, plus the exact
basis tables.  This is what pins the oracle; the GPU parity tests then compare the HIP path with oracle + vectors."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import GOLDEN, golden_cases, load_golden
from helpers import NORMS, oracle_forward, relerr
from oracle import kan_oracle as O


class _Params(nn.Module):
    """Bare parameter holder exposing the reference's attribute names (no reference code involved)."""

    def __init__(self, d):
        super().__init__()
        c = d["cfg"]
        G = c["groups"]
        self.grid_size = c.get("grid_size", 5 if c["kind"] == "bspline" else 8) if c["kind"] != "fourier" else c["degree"]
        self.spline_order = c.get("spline_order", 3)
        self.grid_range = c.get("grid_range", [-1, 1] if c["kind"] == "bspline" else [-2, 2])
        self.degree = c.get("degree", 3)
        names = sorted({k[3:].split(".")[0] for k in d if k.startswith("sd.")})
        for n in names:
            if n in ("base_conv", "spline_conv", "poly_conv", "fourier_conv", "relukan_conv", "prelus"):
                lst = nn.ParameterList([nn.Parameter(torch.from_numpy(d[f"sd.{n}.{g}.weight"])) for g in range(G)])
                setattr(self, n + "_p", lst)
        if c["kind"] == "wav":                                    # Wav-KAN: nested module names, kept verbatim
            self._wav_names = [k[3:] for k in d if k.startswith("sd.") and not k.startswith("sd.layer_norm")]
            self._wav_params = nn.ParameterList([nn.Parameter(torch.from_numpy(d["sd." + n])) for n in self._wav_names])
        if "sd.poly_weights" in d:                                # JacobiKAN: one [G, O/G, C/G*(deg+1), k, k] parameter
            self.poly_weights = nn.Parameter(torch.from_numpy(d["sd.poly_weights"]))
            self.a, self.b = c.get("extra", {}).get("a"), c.get("extra", {}).get("b")
        if "sd.beta_weights" in d:                                # GRAM-KAN: layer-global recurrence parameters
            self.beta_weights = nn.Parameter(torch.from_numpy(d["sd.beta_weights"]))
        if "sd.phase_low" in d:                                   # ReLU-KAN: per-channel phases, g / k from the case
            self.phase_low, self.phase_high = (nn.Parameter(torch.from_numpy(d["sd.phase_" + w])) for w in ("low", "high"))
            self.g, self.k = c.get("extra", {}).get("g", 5), c.get("extra", {}).get("k", 3)
        norm_cls = {1: nn.InstanceNorm1d, 3: nn.InstanceNorm3d}.get(c.get("ndim", 2)) or NORMS[c.get("norm", "in")]
        nch = (c["C"] if c["kind"] == "rbf" else c["O"]) // G
        kw = {k: v for k, v in c.get("norm_kwargs", {}).items()}
        self.layer_norm = nn.ModuleList([norm_cls(nch, **kw) for _ in range(G)])
        for g in range(G):
            for pn in ("weight", "bias"):
                key = f"sd.layer_norm.{g}.{pn}"
                if key in d:
                    getattr(self.layer_norm[g], pn).data.copy_(torch.from_numpy(d[key]))

    def named_parameters(self, *a, **k):       # names as in the reference's state_dict
        for n, p in super().named_parameters(*a, **k):
            if n.startswith("_wav_params."):
                yield self._wav_names[int(n.split(".")[1])], p
            else:
                yield (n.replace("_p.", ".") + ".weight" if "_p." in n else n), p


@pytest.mark.parametrize("name", golden_cases())
def test_oracle_matches_golden(name):
    d = load_golden(name)
    c = d["cfg"]
    holder = _Params(d).train()
    x = torch.from_numpy(d["x"]).requires_grad_(True)
    pre = []
    y = oracle_forward(c, holder, x, pre)
    y.backward(torch.from_numpy(d["g"]))
    assert relerr(y, torch.from_numpy(d["y"])) <= 2e-6
    assert relerr(x.grad, torch.from_numpy(d["dx"])) <= 2e-6
    if "z" in d:
        assert relerr(torch.cat(pre, 1), torch.from_numpy(d["z"])) <= 2e-6
    params = dict(holder.named_parameters())
    for k in d:
        if k.startswith("grad."):
            assert relerr(params[k[5:]].grad, torch.from_numpy(d[k])) <= 5e-6, k


def test_basis_tables_exact():
    d = np.load(os.path.join(GOLDEN, "basis_probes.npz"))
    xs = torch.from_numpy(d["x"])
    for key in [k[:-6] for k in d.files if k.endswith(".knots")]:
        _, G, S, r = key.split("_")
        G, S, r = int(G[1:]), int(S[1:]), float(r[1:])
        knots = O.bspline_knots(G, S, [-r, r])
        assert torch.equal(knots, torch.from_numpy(d[key + ".knots"]))
        assert torch.equal(O.bspline_basis(xs, knots, S), torch.from_numpy(d[key + ".table"]))


def test_known_values():
    # x = 0 -> [0,0,.0208,.4792,.4792,.0208,0,0];  x = -1 -> [1/6, 2/3, 1/6, 0...]
    k = O.bspline_knots(5, 3, [-1, 1])
    b0 = O.bspline_basis(torch.tensor([0.0]), k, 3)[0]
    assert torch.allclose(b0, torch.tensor([0, 0, 1 / 48, 23 / 48, 23 / 48, 1 / 48, 0, 0]), atol=1e-6)
    b1 = O.bspline_basis(torch.tensor([-1.0]), k, 3)[0]
    assert torch.allclose(b1[:3], torch.tensor([1 / 6, 2 / 3, 1 / 6]), atol=1e-6) and float(b1[3:].abs().max()) < 1e-6
    assert float(O.bspline_basis(torch.tensor([2.3, -2.3, float("nan")]), k, 3).abs().nan_to_num().max()) == 0.0
    c, dn = O.rbf_grid(8, [-2, 2])
    assert abs(dn - 4 / 7) < 1e-12 and torch.allclose(O.rbf_basis(torch.tensor([c[3].item()]), c, dn)[0, 3], torch.tensor(1.0))
    t = O.cheby_basis(torch.zeros(1, 1, 1, 1), 4)
    assert torch.allclose(t.flatten(), torch.tensor([1.0, 0.0, -1.0, 0.0, 1.0]), atol=1e-6)
