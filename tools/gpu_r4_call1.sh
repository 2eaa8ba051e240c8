#!/bin/bash
# round 4, call 1: (a) the sharded path's GPU tests with the host-staged exact-size exchange; (b) the five blind variants of round 3
OUT=gpurun_out/r04_call1
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_train_tiny_gpt.py tests/test_gpu_bench_multirank.py -m gpu -q -p no:cacheprovider > $OUT/pytest_sharded.log 2>&1; echo "exit $?" >> $OUT/pytest_sharded.log
tail -15 $OUT/pytest_sharded.log
bash tools/gpu_r4_prepared.sh > $OUT/prepared.log 2>&1
tail -80 $OUT/prepared.log
