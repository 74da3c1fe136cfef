# round 3, call C: band forward (2 taps per barrier step) + band weight gradient: band tests, fuzz + optim + oracle suites, default bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r3c && rm -rf $O && mkdir -p $O &&
(timeout -k 10 300 python -m pytest tests/test_gpu_band.py -q -m gpu > $O/band.txt 2>&1 ; echo "band rc $?" ; tail -4 $O/band.txt) &&
(timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py tests/test_optim.py tests/test_gpu_oracle.py tests/test_gpu_golden.py -q -m gpu > $O/tests.txt 2>&1 ; echo "pytest rc $?" ; tail -4 $O/tests.txt) &&
(timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err ; echo "bench rc $?" ; tail -c 300 $O/bench.json)
