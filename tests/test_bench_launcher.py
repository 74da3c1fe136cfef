"""bench.py --gpus N (N > 1) starts its own ranks: the parent decides from the arguments alone, runs a torch.distributed.run
child on 127.0.0.1, relays exactly one JSON line and the child's exit code.  Exercised here with a gloo stand-in worker."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "_rank_stub.py")


def _launch(extra):
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import bench; "
            f"bench.launch_ranks(2, {extra!r}, script={STUB!r}, check_devices=False)")
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240)


@pytest.mark.timeout(300)
def test_launcher_relays_one_json_line():
    r = _launch(["--spawn"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 2 and doc["value"] == 3.0 and doc["launched"] == "1"
    assert "banner noise" in r.stderr                      # anything else the ranks print on stdout goes to stderr


@pytest.mark.timeout(300)
def test_launcher_propagates_failure():
    r = _launch(["--fail"])
    assert r.returncode != 0
    assert not r.stdout.strip()


def test_direct_multi_gpu_start_without_devices_fails_loudly():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=240,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    if r.returncode == 0:
        pytest.skip("this box has >= 2 GPUs")
    assert "GPU" in (r.stderr + r.stdout)
