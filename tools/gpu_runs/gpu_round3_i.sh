# round 3, call I: generic (run-time P) kernels on the streamed plane expansion: parity of every family, family bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r3i && rm -rf $O && mkdir -p $O &&
(timeout -k 10 700 python -m pytest tests/test_gpu_golden.py tests/test_gpu_fuzz.py tests/test_gpu_oracle.py -q -m gpu > $O/tests.txt 2>&1 ; rc=$? ; echo "pytest rc $rc" ; tail -4 $O/tests.txt ; test $rc -eq 0) &&
(timeout -k 10 400 python tools/family_bench.py > $O/family_bench.txt 2>&1 ; echo "family rc $?" ; cat $O/family_bench.txt)
