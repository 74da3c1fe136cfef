# round 3, call AA: XCD-contiguous order inside the classes of the expanded-operand weight gradient: parity, A/B on one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r3aa && rm -rf $O && mkdir -p $O &&
(timeout -k 10 500 python -m pytest tests/test_gpu_oracle.py tests/test_gpu_models.py -q -m gpu -k "expanded or vgg11 or full_size or deterministic or bs256" > $O/tests.txt 2>&1 ; rc=$? ; echo "pytest rc $rc" ; tail -3 $O/tests.txt ; test $rc -eq 0) &&
(timeout -k 10 400 python tools/ab_env.py KAN_PM_XCD > $O/ab_xcd.txt 2>&1 ; echo "ab rc $?" ; cat $O/ab_xcd.txt | cut -c1-330)
