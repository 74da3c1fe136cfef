# round 3, call D: band kernels after the plan fix: band tests, whole suite, bench, kernel trace of the FastKAN layer, PMC passes of ChebyKAN-AlexNet
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r3d && rm -rf $O && mkdir -p $O &&
(timeout -k 10 300 python -m pytest tests/test_gpu_band.py -q -m gpu > $O/band.txt 2>&1 ; rc=$? ; echo "band rc $rc" ; tail -4 $O/band.txt ; test $rc -eq 0) &&
(timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err ; echo "bench rc $?" ; tail -c 200 $O/bench.json) &&
(timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fk -o fk -- python3 bench.py --workload fastkan_layer --steps 20 --warmup 5 --no-cpu-baseline > $O/fk.log 2>&1 ; echo "fk rc $?") &&
(timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $O/A -o a -- python3 bench.py --workload cheby_alexnet --steps 2 --warmup 1 --no-cpu-baseline > $O/A.log 2>&1 ; echo "pmcA rc $?") &&
(timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVES --output-format csv -d $O/B -o b -- python3 bench.py --workload cheby_alexnet --steps 2 --warmup 1 --no-cpu-baseline > $O/B.log 2>&1 ; echo "pmcB rc $?") &&
rm -f $O/*/*agent_info.csv &&
(timeout -k 10 840 python -m pytest tests -q -m gpu --durations=8 > $O/tests.txt 2>&1 ; echo "pytest rc $?" ; tail -5 $O/tests.txt)
