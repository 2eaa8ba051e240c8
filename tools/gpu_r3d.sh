#!/bin/bash
# Round 3, GPU call D: the three tests that failed in call C, every N-D (mode-product) test on the split mode Gram, a LeNet5 dispatch
# trace on the one-workgroup norm-bound path, and -- bounded -- the 26-dim case.
TAG=${1:-r03d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
timeout 300 python -m pytest tests/test_gpu_kron.py -m gpu -q -k "kwns4_step or t7x5x3 or t4x6x5x3 or lenet" -p no:cacheprovider > $OUT/pytest_a.log 2>&1; echo "exit $?" >> $OUT/pytest_a.log
timeout 300 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "dims and not 26" --durations=8 -p no:cacheprovider > $OUT/pytest_b.log 2>&1; echo "exit $?" >> $OUT/pytest_b.log
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p_l5 -- python $R/bench.py --config lenet5 --steps 12 --warmup 4 --no-cpu-baseline --no-apply-only --no-peaks > $R/$OUT/bench_lenet5_under_rocprof.json 2> $R/$OUT/rocprof.err
  db=$(find /tmp/p_l5 -name "*.db" | head -1); python $R/tools/rocpd_sequence.py $db accumulate_kernel -3 > $R/$OUT/lenet5_step_sequence.md )
timeout 60 python bench.py --config lenet5 --steps 50 --warmup 10 --no-cpu-baseline --no-peaks 2>> $OUT/bench.err | tail -1 > $OUT/bench_lenet5.json
PSGDK_SLOW_TESTS=1 timeout 330 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "dims and 26" --durations=3 -p no:cacheprovider > $OUT/pytest_26.log 2>&1; echo "exit $?" >> $OUT/pytest_26.log
tail -6 $OUT/pytest_a.log; tail -12 $OUT/pytest_b.log; tail -8 $OUT/pytest_26.log; cat $OUT/lenet5_step_sequence.md; head -c 300 $OUT/bench_lenet5.json
