# round 3, call M: ReLU-KAN with a host-applied base activation (two reference fixtures), the phased op with a second input tensor
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r3m && rm -rf $O && mkdir -p $O &&
(timeout -k 10 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_fuzz.py tests/test_gpu_oracle.py -q -m gpu > $O/tests.txt 2>&1 ; rc=$? ; echo "pytest rc $rc" ; tail -5 $O/tests.txt ; test $rc -eq 0)
