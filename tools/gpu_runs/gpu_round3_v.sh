# round 3, call V: ReLU-KAN phase gradients by a reduce-scatter butterfly: parity, family bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r3v && rm -rf $O && mkdir -p $O &&
(timeout -k 10 600 python -m pytest tests/test_gpu_golden.py tests/test_gpu_fuzz.py tests/test_gpu_oracle.py -q -m gpu -k "relu or ReLU" > $O/tests.txt 2>&1 ; rc=$? ; echo "pytest rc $rc" ; tail -3 $O/tests.txt ; test $rc -eq 0) &&
(timeout -k 10 300 python tools/family_bench.py ReLUKAN KAN > $O/family.txt 2>&1 ; echo "rc $?"; tail -2 $O/family.txt)
