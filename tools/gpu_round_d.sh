#!/bin/bash
TAG=${1:-r02d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/gemm_cold.py > $OUT/gemm_cold.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_nlb.py -q -p no:cacheprovider > $OUT/pytest_nlb.log 2>&1
cat $OUT/gemm_cold.txt; tail -15 $OUT/pytest_nlb.log
