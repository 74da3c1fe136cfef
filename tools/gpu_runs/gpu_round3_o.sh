# round 3, call O: kernel stats of ChebyKAN-AlexNet with the fused MaxPool2d(3, 2)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r3o && rm -rf $O && mkdir -p $O &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cheby -o ch -- python3 bench.py --workload cheby_alexnet --steps 5 --warmup 2 --no-cpu-baseline > $O/cheby.log 2>&1 &&
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r3o/cheby/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows:
    n=r['Name']
    if 'in_prelu' in n or 'pool' in n: print(n[:110], r['Calls'], '%.1f us'%(float(r['AverageNs'])/1e3))
PY
