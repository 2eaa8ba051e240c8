#!/bin/bash
# Round 3, GPU call B: the whole GPU suite on the current build, bench lines of every config, a LeNet5 dispatch trace.
TAG=${1:-r03b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
timeout 900 python -m pytest tests -m gpu -q --durations=25 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "exit $?" >> $OUT/pytest_gpu.log
timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
for c in gpt2-medium lenet5 gpt2-small-eq vit-b-lra; do
    timeout 200 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-peaks 2>> $OUT/bench.err | tail -1 > $OUT/bench_$c.json
done
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p_new -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-apply-only --no-peaks > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/rocprof.err
  db=$(find /tmp/p_new -name "*.db" | head -1); python $R/tools/rocpd_stats.py $db > $R/$OUT/kernel_stats.md; python $R/tools/rocpd_sequence.py $db accumulate_kernel -3 > $R/$OUT/step_sequence.md )
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p_l5 -- python $R/bench.py --config lenet5 --steps 12 --warmup 4 --no-cpu-baseline --no-apply-only --no-peaks > /dev/null 2>> $R/$OUT/rocprof.err
  db=$(find /tmp/p_l5 -name "*.db" | head -1); python $R/tools/rocpd_sequence.py $db accumulate_kernel -3 > $R/$OUT/lenet5_step_sequence.md )
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p_med -- python $R/bench.py --config gpt2-medium --steps 8 --warmup 3 --no-cpu-baseline --no-apply-only --no-peaks > /dev/null 2>> $R/$OUT/rocprof.err
  db=$(find /tmp/p_med -name "*.db" | head -1); python $R/tools/rocpd_sequence.py $db accumulate_kernel -3 > $R/$OUT/gpt2-medium_step_sequence.md )
tail -8 $OUT/pytest_gpu.log; head -c 600 $OUT/bench.json; echo; for c in gpt2-medium lenet5 gpt2-small-eq vit-b-lra; do head -c 250 $OUT/bench_$c.json; echo; done
