# round 3, call A: full -m gpu suite on the calibrated tolerances, the exact-transcendentals A/B, the default bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r3a && rm -rf $O && mkdir -p $O &&
(timeout -k 10 840 python -m pytest tests -q -m gpu -x --durations=15 > $O/tests.txt 2>&1 ; echo "pytest rc $?" >> $O/tests.txt ; tail -5 $O/tests.txt) &&
(timeout -k 10 420 python tests/exact_ab.py --out $O/exact_ab.json > $O/exact_ab.log 2>&1 ; echo "exact_ab rc $?" ; tail -3 $O/exact_ab.log) &&
(timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err ; echo "bench rc $?" ; tail -c 600 $O/bench.json)
