# round 3, call L: timing variants of the split-precision prototype (where the time goes); config-5 test
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r3l && rm -rf $O && mkdir -p $O &&
for v in split_bf16_conv sc_nowait sc_nobar sc_nodma sc_pure sc_row90; do
  (timeout -k 10 120 tools/probe/$v 20 > $O/$v.txt 2>&1 ; echo "$v rc $?" ; grep "k_split_fwd" $O/$v.txt) || exit 1
done &&
(timeout -k 10 600 python -m pytest tests/test_gpu_models.py -q -m gpu -k "config5" -s > $O/tests.txt 2>&1 ; rc=$? ; echo "pytest rc $rc" ; grep -E "config 5 full|passed|failed" $O/tests.txt ; test $rc -eq 0)
