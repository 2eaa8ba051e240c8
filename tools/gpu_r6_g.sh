#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r6g; mkdir -p $O
export TMPDIR=/tmp
echo "== default" > $O/fuse_bench.txt
timeout 600 python tools/fuse_bench.py 2>&1 | tail -2 >> $O/fuse_bench.txt
echo "== 128 x 128 kernel everywhere" >> $O/fuse_bench.txt
PSGDK_BIG_MIN_TILES=100000000 timeout 600 python tools/fuse_bench.py 2>&1 | tail -2 >> $O/fuse_bench.txt
cat $O/fuse_bench.txt
