# round 3, call H: band kernels after the vector-ALU diet (magic-number divisions, zero row only, run-time slot count)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r3h && rm -rf $O && mkdir -p $O &&
(timeout -k 10 300 python -m pytest tests/test_gpu_band.py -q -m gpu > $O/band.txt 2>&1 ; rc=$? ; echo "band rc $rc" ; tail -3 $O/band.txt ; test $rc -eq 0) &&
(timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err ; echo "bench rc $?" ; tail -c 200 $O/bench.json) &&
(timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_oracle.py tests/test_gpu_golden.py -q -m gpu > $O/tests.txt 2>&1 ; echo "pytest rc $?" ; tail -4 $O/tests.txt)
