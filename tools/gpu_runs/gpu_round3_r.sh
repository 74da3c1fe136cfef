# round 3, call R: train.GraphedStep (HIP-graph training steps) against eager steps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r3r && rm -rf $O && mkdir -p $O &&
(timeout -k 10 600 python -m pytest tests/test_gpu_models.py -q -m gpu -k "graph" -x > $O/tests.txt 2>&1 ; rc=$? ; echo "pytest rc $rc" ; tail -25 $O/tests.txt ; test $rc -eq 0)
