"""In-tree build of libkanconv.so with hipcc for gfx950 (no JIT cache: the .so travels with the repo)."""
import fcntl
import os
import shutil
import subprocess
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
SOURCES = [os.path.join(_HERE, "csrc", "kanconv.hip")]
HEADERS = [os.path.join(_HERE, "csrc", "kan_device.h"), os.path.join(_HERE, "csrc", "wavkan.inc"), os.path.join(_ROOT, "include", "kanconv.h")]
OUTPUT = os.path.join(_HERE, "libkanconv.so")


def _stale() -> bool:
    if not os.path.exists(OUTPUT):
        return True
    t = os.path.getmtime(OUTPUT)
    return any(os.path.getmtime(f) > t for f in SOURCES + HEADERS)


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/kanconv.hip -> libkanconv.so (skipped when up to date).  Returns the .so path."""
    if not force and not _stale():
        return OUTPUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libkanconv.so")
    # several ranks of one node may get here at once: serialise on a lock file, re-check, and publish atomically
    with open(os.path.join(_HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():
                return OUTPUT
            fd, tmp = tempfile.mkstemp(suffix=".so", dir=_HERE)
            os.close(fd)
            cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-I", os.path.join(_ROOT, "include"),
                   "-I", os.path.join(_HERE, "csrc"), "-o", tmp] + SOURCES
            if os.environ.get("KAN_BUILD_TUNING_KNOBS"):      # experiment builds only: the shipped library reads no environment variable
                cmd.insert(1, "-DKAN_TUNING_KNOBS")
            if verbose:
                print(" ".join(cmd))
            try:
                subprocess.run(cmd, check=True)
                os.replace(tmp, OUTPUT)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return OUTPUT
