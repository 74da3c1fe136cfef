import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def golden_cases(kind=None):
    out = []
    for fn in sorted(os.listdir(GOLDEN)):
        if fn.endswith(".npz") and not fn.startswith(("model_", "basis_")):
            if (kind is None and not fn.startswith("mlp_")) or (kind is not None and fn.startswith(kind + "_")):
                out.append(fn[:-4])
    return out


def load_golden(name):
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    d["cfg"] = json.loads(bytes(d["cfg"]).decode())
    d["noise"] = json.loads(bytes(d["noise"]).decode()) if "noise" in d else {}
    return d


@pytest.fixture(scope="session")
def gpu_lib():
    """Build (if needed) and load libkanconv.so; GPU tests fail -- not skip -- when it is unavailable."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    import convkan_amd
    convkan_amd.build_library()
    return convkan_amd._lib.load()
