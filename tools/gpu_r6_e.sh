#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r6e; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fused_update.py -q 2>&1 | tail -40 > $O/pytest_fused.log
timeout 600 python tools/fuse_bench.py 2>&1 | tail -2 > $O/fuse_bench.txt
tail -n 12 $O/pytest_fused.log; cat $O/fuse_bench.txt
