# round 3, call AB: fabric-side fetch traffic of the expanded-operand weight gradient with and without the XCD-contiguous class order
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r3ab && rm -rf $O && mkdir -p $O &&
export KANCONV_LIB=$GRAFT_REPO_ROOT/convolutional-kan-for-image-classification_amd/libkanconv_kan_tuning_knobs.so &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/on -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-aux > $O/on.log 2>&1 &&
export KAN_PM_XCD=0 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/off -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-aux > $O/off.log 2>&1 &&
python - <<'PY'
import csv, glob, re, collections
for arm in ("off", "on"):
    acc = collections.defaultdict(list)
    for fn in glob.glob(f'gpurun_out/r3ab/{arm}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(fn)):
            m = re.search(r"(k_conv_[a-z_]+)", r["Kernel_Name"])
            if m and r["Counter_Name"] == "FETCH_SIZE": acc[m.group(1)].append(float(r["Counter_Value"]))
    print(arm, {k: round(sum(v) / len(v) / 1024, 1) for k, v in sorted(acc.items())}, "(FETCH_SIZE, MB per launch as counted)")
PY
