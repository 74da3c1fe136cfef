# round 3, call U: kernel breakdown of the generic-kernel families (ReLU-KAN, FourierKAN) on KAN-VGG11
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r3u && rm -rf $O && mkdir -p $O &&
for f in ReLUKAN FourierKAN; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$f -o st -- python3 tools/family_bench.py $f > $O/$f.log 2>&1 || exit 1
  tail -1 $O/$f.log
done
python - <<'PY'
import csv, glob
for fam in ("ReLUKAN", "FourierKAN"):
    f = glob.glob(f'gpurun_out/r3u/{fam}/**/*kernel_stats.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print(fam, 'total ms over 13 steps', tot / 1e6)
    for r in rows[:12]:
        print('  ', r['Name'][:100].ljust(100), r['Calls'].rjust(5), '%8.1f us' % (float(r['AverageNs']) / 1e3), '%5.1f%%' % (100 * float(r['TotalDurationNs']) / tot))
PY
