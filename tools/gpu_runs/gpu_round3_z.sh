cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r3z && rm -rf $O && mkdir -p $O &&
(timeout -k 10 600 python -m pytest tests/test_gpu_pool.py -q -m gpu > $O/tests.txt 2>&1 ; rc=$? ; echo "pytest rc $rc" ; tail -12 $O/tests.txt ; test $rc -eq 0)
