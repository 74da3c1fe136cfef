cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r3s && rm -rf $O && mkdir -p $O &&
(timeout -k 10 600 python tools/graph_bench.py KAN 4 16 64 256 > $O/graph_bench.txt 2>&1 ; echo "rc $?"; cat $O/graph_bench.txt) &&
(timeout -k 10 600 python tools/graph_bench.py ChebyKAN 16 256 >> $O/graph_bench.txt 2>&1 ; echo "rc $?"; tail -2 $O/graph_bench.txt)
