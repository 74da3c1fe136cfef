# round 3, call G: longest-first dispatch order of the expanded-operand weight gradient: parity on the small-plane tests, A/B on one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r3g && rm -rf $O && mkdir -p $O &&
(timeout -k 10 400 python -m pytest tests/test_gpu_oracle.py tests/test_gpu_models.py -q -m gpu -k "expanded or vgg11 or full_size or deterministic or bs256" > $O/tests.txt 2>&1 ; rc=$? ; echo "pytest rc $rc" ; tail -3 $O/tests.txt ; test $rc -eq 0) &&
(timeout -k 10 400 python tools/ab_env.py KAN_PM_LPT > $O/ab_lpt.txt 2>&1 ; echo "ab rc $?" ; cat $O/ab_lpt.txt | cut -c1-400)
