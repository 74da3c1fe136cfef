# Round profile set (default bench, kernel-trace stats, four PMC passes, the two auxiliary workloads) under gpurun_out/prof3; summarise with tools/pmc_report.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/prof3 && rm -rf $O && mkdir -p $O &&
python bench.py > $O/bench_default.json 2> $O/bench_default.err &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-aux > $O/stats.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $O/A -o a -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-aux > $O/A.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVES --output-format csv -d $O/B -o b -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-aux > $O/B.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/F -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-aux > $O/F.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/W -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-aux > $O/W.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cheby -o ch -- python3 bench.py --workload cheby_alexnet --steps 5 --warmup 2 --no-cpu-baseline > $O/cheby.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fk -o fk -- python3 bench.py --workload fastkan_layer --steps 20 --warmup 5 --no-cpu-baseline > $O/fk.log 2>&1 &&
rm -f $O/*/*agent_info.csv && ls $O && tail -c 300 $O/bench_default.json
