#!/bin/bash
# Round 3, GPU call G: the 8-wide mapping of the streaming kernels: parity suites, bench, dispatch trace.
TAG=${1:-r03g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
timeout 600 python -m pytest tests/test_gpu_kron.py tests/test_gpu_production_path.py tests/test_gpu_eq.py tests/test_gpu_fuzz.py tests/test_gpu_sharded.py \
   tests/test_gpu_train_tiny_gpt.py tests/test_gpu_dtensor.py -m gpu -q --durations=5 -p no:cacheprovider > $OUT/pytest_a.log 2>&1; echo "exit $?" >> $OUT/pytest_a.log
timeout 200 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "not lra" -p no:cacheprovider > $OUT/pytest_b.log 2>&1; echo "exit $?" >> $OUT/pytest_b.log
timeout 100 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-peaks > $OUT/bench.json 2> $OUT/bench.err
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p_new -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-apply-only --no-peaks > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/rocprof.err
  db=$(find /tmp/p_new -name "*.db" | head -1); python $R/tools/rocpd_sequence.py $db accumulate_kernel -3 > $R/$OUT/step_sequence.md )
tail -9 $OUT/pytest_a.log; tail -4 $OUT/pytest_b.log; head -c 260 $OUT/bench.json; echo; grep "accumulate\|emit" $OUT/step_sequence.md; tail -1 $OUT/step_sequence.md
