# round 3, call N: fused MaxPool2d(k, s) in the norm kernels: parity, AlexNet model tests, AlexNet step time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r3n && rm -rf $O && mkdir -p $O &&
(timeout -k 10 900 python -m pytest tests/test_gpu_pool.py tests/test_gpu_models.py -q -m gpu -x > $O/tests.txt 2>&1 ; rc=$? ; echo "pytest rc $rc" ; tail -15 $O/tests.txt ; test $rc -eq 0) &&
(timeout -k 10 300 python bench.py --workload cheby_alexnet --steps 10 --warmup 3 --no-cpu-baseline > $O/cheby.json 2> $O/cheby.err ; echo "bench rc $?"; python -c "
import json,sys; d=json.loads(open('$O/cheby.json').read().strip().splitlines()[-1]); print(d.get('ms_per_step'), d.get('value'))") &&
(KAN_FUSE_POOL=0 timeout -k 10 300 python bench.py --workload cheby_alexnet --steps 10 --warmup 3 --no-cpu-baseline > $O/cheby_unfused.json 2> $O/cheby_unfused.err ; echo "bench rc $?"; python -c "
import json,sys; d=json.loads(open('$O/cheby_unfused.json').read().strip().splitlines()[-1]); print('unfused', d.get('ms_per_step'), d.get('value'))")
