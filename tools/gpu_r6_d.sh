#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r6d; mkdir -p $O
export TMPDIR=/tmp
cp psgd_torch_amd/libpsgdk.so /tmp/lib_keep.so
for lib in nt plainst plain nt; do
  cp ab_libs/lib_$lib.so psgd_torch_amd/libpsgdk.so
  echo "== $lib" >> $O/fuse_bench.txt
  timeout 600 python tools/fuse_bench.py 2>&1 | tail -2 >> $O/fuse_bench.txt
done
cp /tmp/lib_keep.so psgd_torch_amd/libpsgdk.so
timeout 900 python -m pytest tests/test_gpu_fused_update.py -x -q 2>&1 | tail -25 > $O/pytest_fused.log
tail -n 4 $O/pytest_fused.log; cat $O/fuse_bench.txt
