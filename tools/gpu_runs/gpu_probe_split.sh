# one run of the split-precision prototype (tools/probe/split_bf16_conv); extra binaries sc_* if present
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/split && mkdir -p $O &&
for v in split_bf16_conv $(cd tools/probe && ls sc_* 2>/dev/null); do
  (timeout -k 10 120 tools/probe/$v 20 > $O/$v.txt 2>&1 ; echo "$v rc $?" ; grep -E "k_split_fwd|kan_conv_fwd|clock" $O/$v.txt) || exit 1
done
