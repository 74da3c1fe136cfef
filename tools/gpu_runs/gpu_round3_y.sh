cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r3y && rm -rf $O && mkdir -p $O &&
(timeout -k 10 600 python -m pytest tests/test_gpu_split.py -q -m gpu -s > $O/tests.txt 2>&1 ; rc=$? ; echo "pytest rc $rc" ; grep -E "split|passed|failed|Error" $O/tests.txt | tail -12 ; test $rc -eq 0) &&
(timeout -k 10 300 python tools/infer_bench.py > $O/infer.txt 2>&1 ; echo "rc $?"; tail -1 $O/infer.txt)
