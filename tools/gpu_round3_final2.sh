# round 3, final evidence 2/3: exact-transcendentals A/B, noise probe, split-precision probes, LPT A/B, family bench, repeats of the default bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/fin3b && rm -rf $O && mkdir -p $O &&
(timeout -k 10 420 python tests/exact_ab.py --out $O/exact_ab.json > $O/exact_ab.log 2>&1 ; echo "exact_ab rc $?" ; tail -2 $O/exact_ab.log) &&
(timeout -k 10 200 python tests/noise_probe.py > $O/noise_probe.txt 2>&1 ; echo "noise rc $?") &&
(timeout -k 10 120 ./tools/probe/split_bf16_probe > $O/split_bf16_probe.txt 2>&1 ; echo "split probe rc $?") &&
(timeout -k 10 120 ./tools/probe/split_bf16_conv 20 > $O/split_bf16_conv.txt 2>&1 ; echo "split conv rc $?"; cat $O/split_bf16_conv.txt) &&
(timeout -k 10 120 ./tools/probe/sc_stamp 200 > $O/split_bf16_conv_stamp.txt 2>&1 ; echo "split stamp rc $?"; grep clock $O/split_bf16_conv_stamp.txt) &&
(timeout -k 10 400 python tools/ab_env.py KAN_PM_LPT > $O/ab_lpt.txt 2>&1 ; echo "ab rc $?" ; tail -1 $O/ab_lpt.txt) &&
(timeout -k 10 400 python tools/family_bench.py > $O/family_bench.txt 2>&1 ; echo "family rc $?" ; cat $O/family_bench.txt) &&
(for i in 1 2 3 4 5; do timeout -k 10 120 python bench.py --no-cpu-baseline --no-aux 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" ; done > $O/repeats.txt ; cat $O/repeats.txt)
