# Full GPU test suite on the box (output under gpurun_out/fin): /usr/local/graft/bin/gpurun --timeout 1100 -- "bash tools/gpu_suite.sh"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/fin &&
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/fin/tests.txt 2>&1 ; tail -3 gpurun_out/fin/tests.txt
