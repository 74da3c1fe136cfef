# round 3, call P: opt-in split-precision forward in the library: parity tests, bench aux entry
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r3p && rm -rf $O && mkdir -p $O &&
(timeout -k 10 600 python -m pytest tests/test_gpu_split.py -q -m gpu -s > $O/tests.txt 2>&1 ; rc=$? ; echo "pytest rc $rc" ; grep -E "split |passed|failed|Error" $O/tests.txt | head -20 ; test $rc -eq 0) &&
(timeout -k 10 120 python -c "
import json, torch, bench
print(json.dumps(bench.split_precision_forward(torch.device('cuda:0')), indent=1))" > $O/split_bench.txt 2>&1 ; cat $O/split_bench.txt)
