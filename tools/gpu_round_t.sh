#!/bin/bash
TAG=${1:-r02t}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_lra.py tests/test_gpu_nlb.py tests/test_gpu_sharded.py tests/test_gpu_bench_multirank.py tests/test_gpu_dtensor.py -q -p no:cacheprovider > $OUT/pytest.log 2>&1
timeout 300 python bench.py --config vit-b-lra --steps 10 --warmup 3 > $OUT/bench_lra.json 2> $OUT/bench_lra.err
tail -15 $OUT/pytest.log; cat $OUT/bench_lra.json | head -c 900
