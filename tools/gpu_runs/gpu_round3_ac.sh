cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r3ac && rm -rf $O && mkdir -p $O &&
timeout -k 10 300 python - > $O/host.txt 2>&1 <<'PY'
import cProfile, pstats, io, torch, bench
bench._load_torch()
dev = torch.device("cuda:0")
m = bench.build_model(dev, "fastkan_layer")
x = torch.randn(256, 3, 32, 32, device=dev)
for _ in range(20): bench.one_step(m, x, None)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): bench.one_step(m, x, None)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
PY
tail -45 $O/host.txt
