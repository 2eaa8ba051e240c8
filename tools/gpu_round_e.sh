#!/bin/bash
TAG=${1:-r02e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kron.py tests/test_gpu_fullsize.py -q -p no:cacheprovider -k "not lra" > $OUT/pytest.log 2>&1
timeout 600 python tools/gemm_cold.py > $OUT/gemm_cold.txt 2>&1
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_pipe.json 2> $OUT/bench_pipe.err
PSGDK_BIG_MIN_TILES=500 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_pipe_min500.json 2> $OUT/bench_min500.err
timeout 300 python bench.py --config gpt2-medium --steps 20 --warmup 3 > $OUT/bench_medium.json 2> $OUT/bench_medium.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_stats -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-apply-only > $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err
db=$(find /tmp/p_stats -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_sequence.py $db accumulate_kernel -3 > $GRAFT_REPO_ROOT/$OUT/step_sequence.md
cd $GRAFT_REPO_ROOT
tail -8 $OUT/pytest.log; cat $OUT/gemm_cold.txt; grep "gemm_nt" $OUT/step_sequence.md; tail -2 $OUT/step_sequence.md
for f in $OUT/bench_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); r=d.get('roofline',{})
print(d['ms_per_step'], d['value'], r.get('frac'), r.get('gemm_ms_per_step'))"; done
