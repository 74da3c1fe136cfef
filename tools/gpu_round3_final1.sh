# round 3, final evidence 1/3: whole -m gpu suite, smoke
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/fin3 && rm -rf $O && mkdir -p $O &&
(timeout -k 10 1100 python -m pytest tests -q -m gpu --durations=8 > $O/tests.txt 2>&1 ; rc=$? ; echo "pytest rc $rc" ; tail -4 $O/tests.txt ; test $rc -eq 0) &&
(timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 ; echo "smoke rc $?" ; tail -2 $O/smoke.txt)
