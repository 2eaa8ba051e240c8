#!/bin/bash
# Round 3, GPU call E: start block from one load of A[j], rsub_t with packed LDS tiles, 7-round Philox for the per-element noise:
# the parity suites that cover them, then dispatch traces of GPT-2-small and LeNet5.
TAG=${1:-r03e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
timeout 500 python -m pytest tests/test_gpu_kron.py tests/test_gpu_nlb.py tests/test_gpu_production_path.py tests/test_gpu_eq.py tests/test_gpu_sharded.py \
   tests/test_gpu_train_tiny_gpt.py tests/test_gpu_c_abi.py -m gpu -q --durations=6 -p no:cacheprovider > $OUT/pytest_a.log 2>&1; echo "exit $?" >> $OUT/pytest_a.log
timeout 200 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "not lra" --durations=5 -p no:cacheprovider > $OUT/pytest_b.log 2>&1; echo "exit $?" >> $OUT/pytest_b.log
timeout 100 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-peaks > $OUT/bench.json 2> $OUT/bench.err
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p_new -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-apply-only --no-peaks > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/rocprof.err
  db=$(find /tmp/p_new -name "*.db" | head -1); python $R/tools/rocpd_sequence.py $db accumulate_kernel -3 > $R/$OUT/step_sequence.md )
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p_l5 -- python $R/bench.py --config lenet5 --steps 12 --warmup 4 --no-cpu-baseline --no-apply-only --no-peaks > /dev/null 2>> $R/$OUT/rocprof.err
  db=$(find /tmp/p_l5 -name "*.db" | head -1); python $R/tools/rocpd_sequence.py $db accumulate_kernel -3 > $R/$OUT/lenet5_step_sequence.md )
tail -12 $OUT/pytest_a.log; tail -8 $OUT/pytest_b.log; head -c 260 $OUT/bench.json; echo; cat $OUT/step_sequence.md; grep nlb_coop $OUT/lenet5_step_sequence.md
