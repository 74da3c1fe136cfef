# round 3, final evidence 1/2: whole -m gpu suite, smoke, exact-transcendentals A/B, noise probe, split-precision probe, repeats of the default bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/fin3 && rm -rf $O && mkdir -p $O &&
(timeout -k 10 900 python -m pytest tests -q -m gpu --durations=8 > $O/tests.txt 2>&1 ; rc=$? ; echo "pytest rc $rc" ; tail -4 $O/tests.txt ; test $rc -eq 0) &&
(timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 ; echo "smoke rc $?" ; tail -2 $O/smoke.txt) &&
(timeout -k 10 420 python tests/exact_ab.py --out $O/exact_ab.json > $O/exact_ab.log 2>&1 ; echo "exact_ab rc $?" ; tail -2 $O/exact_ab.log) &&
(timeout -k 10 200 python tests/noise_probe.py > $O/noise_probe.txt 2>&1 ; echo "noise rc $?") &&
(timeout -k 10 120 ./tools/probe/split_bf16_probe > $O/split_bf16_probe.txt 2>&1 ; echo "split rc $?") &&
(for i in 1 2 3 4 5; do timeout -k 10 120 python bench.py --no-cpu-baseline --no-aux 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" ; done > $O/repeats.txt ; cat $O/repeats.txt)
