"""bench.py --gpus N (N > 1) starts its own ranks: the parent decides from the arguments alone, runs a torch.distributed.run
child on 127.0.0.1, relays exactly one JSON line and the child's exit code.  Exercised here with a gloo stand-in worker."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "_rank_stub.py")


def _launch(extra, check_devices=False):
    # the pre_popen hook dumps the parent's loaded GPU runtime libraries (from /proc/self/maps) to stderr just before it starts the ranks
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import bench; "
            "hook = lambda: sys.stderr.write('PARENT-GPU-LIBS:' + ','.join(sorted({ln.split('/')[-1].strip() for ln in open('/proc/self/maps') "
            "if any(k in ln for k in ('libamdhip64', 'libhsa-runtime64', 'libtorch', 'librccl'))})) + '\\n'); "
            f"bench.kfd_gpu_count = (lambda: 8) if {check_devices!r} else bench.kfd_gpu_count; "
            f"bench.launch_ranks(2, {extra!r}, script={STUB!r}, check_devices={check_devices!r}, pre_popen=hook)")
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
def test_launcher_parent_never_loads_the_gpu_runtime():
    """An 8-rank node must not have a 9th process holding every device open: at Popen time the parent has neither torch nor the HIP / HSA
    runtime mapped (it counts GPUs in the KFD sysfs topology, stubbed to 8 here so that the device check itself runs)."""
    r = _launch(["--spawn"], check_devices=True)
    assert r.returncode == 0, r.stderr[-2000:]
    marks = [ln for ln in r.stderr.splitlines() if ln.startswith("PARENT-GPU-LIBS:")]
    assert marks == ["PARENT-GPU-LIBS:"], marks


def test_kfd_gpu_count_reads_sysfs_only(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    n = bench.kfd_gpu_count()
    assert n is None or n >= 0
    assert bench.torch is None, "importing bench / counting GPUs must not import torch"


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
