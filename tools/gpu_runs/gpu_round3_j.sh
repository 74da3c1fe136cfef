# round 3, call J: config-5 full-batch test, GRAM parameter gradients on the streamed planes, kernel breakdown of the generic-kernel families
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r3j && rm -rf $O && mkdir -p $O &&
(timeout -k 10 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_golden.py -q -m gpu -k "config5 or gram or relu or cheby_alexnet" -s > $O/tests.txt 2>&1 ; rc=$? ; echo "pytest rc $rc" ; grep -E "config 5 full|passed|failed" $O/tests.txt ; test $rc -eq 0) &&
for f in FourierKAN ReLUKAN WavKAN; do
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/$f -o st -- python3 $GRAFT_REPO_ROOT/tools/family_bench.py $f > $GRAFT_REPO_ROOT/$O/$f.log 2>&1 ; echo "$f rc $?"; tail -1 $GRAFT_REPO_ROOT/$O/$f.log) || exit 1
done
