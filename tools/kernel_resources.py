#!/usr/bin/env python3
"""Register / LDS / spill figures of every kernel in libkanconv.so (from the gfx950 code object's metadata notes).
usage: kernel_resources.py [substring]"""
import os, re, struct, subprocess, sys, tempfile
so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "convolutional-kan-for-image-classification_amd", "libkanconv.so")
with tempfile.TemporaryDirectory() as d:
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, d + "/fat.bin"], check=True)
    blob = open(d + "/fat.bin", "rb").read()
    notes, start, k = "", 0, 0
    while True:                                   # one offload bundle per translation unit
        start = blob.find(b"__CLANG_OFFLOAD_BUNDLE__", start)
        if start < 0:
            break
        n, off = struct.unpack("<Q", blob[start + 24:start + 32])[0], start + 32
        for _ in range(n):
            o, sz, tl = struct.unpack("<QQQ", blob[off:off + 24]); off += 24
            triple = blob[off:off + tl].decode(); off += tl
            if "gfx950" in triple:
                open(d + f"/dev{k}.co", "wb").write(blob[start + o:start + o + sz])
                notes += subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", d + f"/dev{k}.co"], capture_output=True, text=True).stdout
                k += 1
        start += 24
    demangle = lambda s: subprocess.run(["c++filt", s], capture_output=True, text=True).stdout.strip()
for blk in notes.split("- .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
    short = re.sub(r"\(anonymous namespace\)::|\(.*", "", demangle(name))
    if len(sys.argv) > 1 and sys.argv[1] not in short:
        continue
    print(f"{short:48s} vgpr {g('vgpr_count'):3d} sgpr {g('sgpr_count'):3d} vgpr-spill {g('vgpr_spill_count'):3d} sgpr-spill {g('sgpr_spill_count'):3d} "
          f"lds {g('group_segment_fixed_size'):6d} scratch {g('private_segment_fixed_size'):4d}")
