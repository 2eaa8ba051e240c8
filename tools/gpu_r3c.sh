#!/bin/bash
# Round 3, GPU call C: the whole GPU suite (files not yet run on this build first), then short bench lines.
TAG=${1:-r03c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1050 python -m pytest tests/test_gpu_kron.py tests/test_gpu_lra.py tests/test_gpu_nlb.py tests/test_gpu_sharded.py tests/test_gpu_train_tiny_gpt.py \
    tests/test_gpu_fuzz.py tests/test_gpu_c_abi.py tests/test_gpu_dtensor.py tests/test_gpu_production_path.py tests/test_gpu_eq.py tests/test_gpu_fullsize.py \
    tests/test_gpu_bench_multirank.py -m gpu -q --durations=30 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "exit $?" >> $OUT/pytest_gpu.log
timeout 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-peaks > $OUT/bench.json 2> $OUT/bench.err
timeout 120 python bench.py --config lenet5 --steps 30 --warmup 8 --no-cpu-baseline --no-peaks 2>> $OUT/bench.err | tail -1 > $OUT/bench_lenet5.json
timeout 120 python bench.py --config vit-b-lra --steps 20 --warmup 5 --no-cpu-baseline 2>> $OUT/bench.err | tail -1 > $OUT/bench_vit-b-lra.json
tail -45 $OUT/pytest_gpu.log; head -c 300 $OUT/bench.json; echo; head -c 300 $OUT/bench_lenet5.json; echo; head -c 300 $OUT/bench_vit-b-lra.json
