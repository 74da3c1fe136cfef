# round 3, call T: effective clock under the conv kernels (GRBM_GUI_ACTIVE / 8 / launch duration, MI355X_MICROARCH.md 'DVFS give-back')
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r3t && rm -rf $O && mkdir -p $O &&
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/G -o g -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-aux > $O/G.log 2>&1 &&
python - <<'PY'
import csv, glob, re, collections
acc = collections.defaultdict(list)
for fn in glob.glob('gpurun_out/r3t/G/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        m = re.search(r"(k_(?:conv|band)_[a-z_]+(?:<[^>]*>)?)", r["Kernel_Name"])
        if m and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            ns = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            acc[m.group(1)].append((float(r["Counter_Value"]) / 8.0 / ns, ns))
for k, v in sorted(acc.items()):
    print(f"{k:44s} launches {len(v):3d}  avg {sum(n for _, n in v) / len(v) / 1e3:8.1f} us  effective clock {sum(c for c, _ in v) / len(v):.3f} GHz")
PY
