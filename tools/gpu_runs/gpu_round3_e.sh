# round 3, call E: band forward with 96 x 64 wave tiles (192 outputs) and 256-pixel tiles (big halos); split-precision probe; plane windows
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r3e && rm -rf $O && mkdir -p $O &&
(timeout -k 10 300 python -m pytest tests/test_gpu_band.py tests/test_gpu_golden.py -q -m gpu > $O/band.txt 2>&1 ; rc=$? ; echo "band+golden rc $rc" ; tail -4 $O/band.txt ; test $rc -eq 0) &&
(timeout -k 10 120 ./tools/probe/split_bf16_probe > $O/split_bf16_probe.txt 2>&1 ; echo "probe rc $?" ; cat $O/split_bf16_probe.txt) &&
(timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err ; echo "bench rc $?" ; tail -c 200 $O/bench.json) &&
(timeout -k 10 400 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_oracle.py -q -m gpu > $O/tests.txt 2>&1 ; echo "pytest rc $?" ; tail -4 $O/tests.txt)
