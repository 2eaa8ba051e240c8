#!/bin/bash
TAG=${1:-full}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --durations=12 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "exit $?" >> $OUT/pytest_gpu.log
tail -25 $OUT/pytest_gpu.log
