#!/usr/bin/env python3
"""Build profiles/rNN_hbm_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
usage: hbm_traffic.py <fetch_dir> <write_dir> <out.json>   (dirs hold *counter_collection.csv)"""
import collections, csv, glob, json, re, sys


def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] != counter:
                continue
            m = re.search(r"(k_conv_[a-z_]+)", r["Kernel_Name"])
            if m:
                acc[m.group(1)].append(float(r["Counter_Value"]))
    return acc


f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over `python3 bench.py --steps 2 "
               "--warmup 1 --no-cpu-baseline`, mean per launch over all launches of the kernel family in the run (all 8 KAN-VGG11 "
               "layers, bs=256). Correction per MI355X_MICROARCH.md section HBM: bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (FETCH_SIZE "
               "under-reports wide streaming reads by 2x on gfx950; our 4-B gathers are uncalibrated, so this is an upper estimate of "
               "the read side; Infinity-Cache hits are counted).", "kernels": {}}
for k in sorted(f):
    fk, wk = sum(f[k]) / len(f[k]), sum(w[k]) / max(1, len(w[k]))
    out["kernels"][k] = {"FETCH_SIZE_KB_per_launch": fk, "launches_profiled": len(f[k]), "WRITE_SIZE_KB_per_launch": wk,
                         "hbm_bytes_per_launch_corrected": (2 * fk + wk) * 1024}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
