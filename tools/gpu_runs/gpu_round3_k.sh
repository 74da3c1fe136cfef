# round 3, call K: split-precision forward prototype on the real 256->256@8x8 shape; chunk-consistency diagnostic of the AlexNet layer shapes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && O=gpurun_out/r3k && rm -rf $O && mkdir -p $O &&
(timeout -k 10 120 tools/probe/split_bf16_conv 20 > $O/split_conv.txt 2>&1 ; echo "split rc $?" ; cat $O/split_conv.txt) &&
(timeout -k 10 400 python tools/probe/chunk_consistency.py 128 > $O/chunks.txt 2>&1 ; echo "chunks rc $?" ; cat $O/chunks.txt)
