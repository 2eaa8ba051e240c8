#!/bin/bash
TAG=${1:-r02l}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kron.py tests/test_gpu_eq.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -q -p no:cacheprovider -x -k "not lra" > $OUT/pytest.log 2>&1
timeout 600 python tools/stage_bench.py > $OUT/stage_bench.txt 2>&1
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
timeout 300 python bench.py --config gpt2-medium --steps 20 --warmup 3 > $OUT/bench_medium.json 2> $OUT/bench_medium.err
tail -3 $OUT/pytest.log; cat $OUT/stage_bench.txt | tail -8
for f in $OUT/bench.json $OUT/bench_medium.json; do python -c "
import json,sys
d=json.load(open('$f')); r=d.get('roofline',{})
print(d['ms_per_step'], d['value'], r.get('frac'), r.get('gemm_ms_per_step'))"; done
