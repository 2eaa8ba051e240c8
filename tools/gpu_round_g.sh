#!/bin/bash
TAG=${1:-r02g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/gemm_cold.py > $OUT/gemm_cold.txt 2>&1
cat $OUT/gemm_cold.txt
