#!/bin/bash
TAG=${1:-r02j}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/stage_bench.py > $OUT/stage_bench.txt 2>&1

cat $OUT/stage_bench.txt $OUT/stage_bench_medium.txt
