cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r3x && rm -rf $O && mkdir -p $O &&
(timeout -k 10 700 python -m pytest tests/test_gpu_dp.py -q -m gpu -x > $O/tests.txt 2>&1 ; rc=$? ; echo "pytest rc $rc" ; tail -25 $O/tests.txt ; test $rc -eq 0)
