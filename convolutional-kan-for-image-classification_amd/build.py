"""In-tree build of libkanconv.so with hipcc for gfx950 (no JIT cache: the .so travels with the repo)."""
import fcntl
import os
import shutil
import subprocess
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_CSRC = os.path.join(_HERE, "csrc")
SOURCES = sorted(os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith(".hip"))          # one object per translation unit
HEADERS = sorted(os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".h", ".inc"))) + [os.path.join(_ROOT, "include", "kanconv.h")]
_OBJ = os.path.join(_HERE, "_obj")                # per-unit objects of the shipped build (git-ignored): an edit recompiles its own unit only
OUTPUT = os.path.join(_HERE, "libkanconv.so")


def _stale() -> bool:
    if not os.path.exists(OUTPUT):
        return True
    t = os.path.getmtime(OUTPUT)
    return any(os.path.getmtime(f) > t for f in SOURCES + HEADERS)


def build_library(force: bool = False, verbose: bool = False, defines=(), output: str = None) -> str:
    """Compile csrc/kanconv.hip -> libkanconv.so (skipped when up to date).  Returns the .so path.
    `defines` / `output`: measurement variants only (e.g. ("KAN_EXACT_TRANSCENDENTALS",) -> libkanconv_exact.so for tools/exact_ab.py);
    a variant never replaces the shipped library."""
    if defines or output:
        out = output or os.path.join(_HERE, "libkanconv_" + "_".join(d.lower() for d in defines) + ".so")
        if not force and os.path.exists(out) and os.path.getmtime(out) > max(os.path.getmtime(f) for f in SOURCES + HEADERS):
            return out                                # built in the container, travelled with the snapshot
        return _compile(list(defines), out, verbose)
    if not force and not _stale():
        return OUTPUT
    # several ranks of one node may get here at once: serialise on a lock file, re-check, and publish atomically
    with open(os.path.join(_HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():
                return OUTPUT
            _compile(["KAN_TUNING_KNOBS"] if os.environ.get("KAN_BUILD_TUNING_KNOBS") else [], OUTPUT, verbose)   # knobs: experiment builds only,
        finally:                                                                                               # the shipped library reads no environment variable
            fcntl.flock(lock, fcntl.LOCK_UN)
    return OUTPUT


def _compile(defines, output: str, verbose: bool = False) -> str:
    """hipcc -c every translation unit (in parallel; objects of the plain build are kept and reused while their sources are older),
    then link the shared object and publish it atomically."""
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build " + os.path.basename(output))
    variant = bool(defines) or os.path.abspath(output) != OUTPUT
    objdir = tempfile.mkdtemp(dir=_HERE, prefix="_obj_") if variant else _OBJ
    os.makedirs(objdir, exist_ok=True)
    flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC"] + ["-D" + d for d in defines] + ["-I", os.path.join(_ROOT, "include"), "-I", _CSRC]
    newest_header = max(os.path.getmtime(h) for h in HEADERS)

    def unit(src):
        obj = os.path.join(objdir, os.path.splitext(os.path.basename(src))[0] + ".o")
        if not variant and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), newest_header):
            return obj
        cmd = [hipcc] + flags + ["-c", src, "-o", obj + ".tmp"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        os.replace(obj + ".tmp", obj)
        return obj
    fd, tmp = tempfile.mkstemp(suffix=".so", dir=_HERE)
    os.close(fd)
    try:
        with ThreadPoolExecutor(max_workers=min(4, len(SOURCES))) as ex:
            objs = list(ex.map(unit, SOURCES))
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        os.replace(tmp, output)                   # atomic publish
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
        if variant:
            shutil.rmtree(objdir, ignore_errors=True)
    return output
