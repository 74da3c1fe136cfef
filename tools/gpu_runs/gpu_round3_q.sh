# round 3, call Q: the single-layer FastKAN workload as a HIP graph
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r3q && rm -rf $O && mkdir -p $O &&
(timeout -k 10 300 python -c "
import json, torch, bench
bench._load_torch()
d = bench.other_workload('fastkan_layer', torch.device('cuda:0'), 30, 10)
print(d['ms_per_step'], d.get('hip_graph'), d.get('error'))" > $O/fk.txt 2>&1 ; echo "rc $?"; tail -5 $O/fk.txt)
