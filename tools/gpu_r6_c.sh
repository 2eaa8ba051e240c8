#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r6c; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fused_update.py -x -q 2>&1 | tail -25 > $O/pytest_fused.log
timeout 600 python tools/fuse_bench.py > $O/fuse_bench.txt 2>&1
tail -n 4 $O/pytest_fused.log; cat $O/fuse_bench.txt | tail -5
