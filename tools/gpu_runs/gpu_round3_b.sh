# round 3, call B: band forward tests first, then the whole -m gpu suite (no -x), the noise probe, the default bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r3b && rm -rf $O && mkdir -p $O &&
(timeout -k 10 300 python -m pytest tests/test_gpu_band.py -q -m gpu > $O/band.txt 2>&1 ; echo "band rc $?" ; tail -4 $O/band.txt) &&
(timeout -k 10 200 python tests/noise_probe.py > $O/noise_probe.txt 2>&1 ; echo "probe rc $?") &&
(timeout -k 10 840 python -m pytest tests -q -m gpu --durations=12 > $O/tests.txt 2>&1 ; echo "pytest rc $?" ; tail -4 $O/tests.txt) &&
(timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err ; echo "bench rc $?" ; tail -c 300 $O/bench.json)
