#!/bin/bash
TAG=${1:-r02b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_nlb.py tests/test_gpu_dtensor.py "tests/test_gpu_kron.py::test_gemm_kernel" -q -p no:cacheprovider -x > $OUT/pytest_fix.log 2>&1
echo "exit $?" >> $OUT/pytest_fix.log
timeout 900 python tools/gemm_ab.py > $OUT/gemm_ab.txt 2>&1
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_pipe.json 2> $OUT/bench_pipe.err
PSGDK_GEMM_BIG=lock timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_lock.json 2> $OUT/bench_lock.err
PSGDK_BIG_MIN_TILES=500 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_pipe_min500.json 2> $OUT/bench_min500.err
timeout 300 python bench.py --config gpt2-medium --steps 20 --warmup 3 > $OUT/bench_medium_pipe.json 2> $OUT/bench_medium.err
tail -4 $OUT/pytest_fix.log; cat $OUT/gemm_ab.txt | tail -40
for f in $OUT/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); r=d.get('roofline',{})
print(d['ms_per_step'], d['value'], r.get('frac'), r.get('gemm_ms_per_step'))"; done
