#!/bin/bash
# round 4, call 10: one rank's dispatch sequence of an 8-rank sharded step (GPT-2-small; 1 chunk and the default 2), and the sharded GPU tests on the new default
OUT=$(pwd)/gpurun_out/r04_c10
R=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in 1 2; do
  rm -rf /tmp/p_r$c
  timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p_r$c -- python $R/tools/rank_arithmetic.py --world 8 --rank 4 --chunks $c --steps 12 --warmup 4 > $OUT/rank4_c$c.out 2> $OUT/rank4_c$c.err
  db=$(find /tmp/p_r$c -name "*.db" | head -1)
  python $R/tools/rocpd_sequence.py $db accumulate_kernel $(( -1 - 2 * c )) > $OUT/rank4_of_8_c${c}_step_sequence.md
  python $R/tools/rocpd_stats.py $db > $OUT/rank4_of_8_c${c}_kernel_stats.md
  cat $OUT/rank4_of_8_c${c}_step_sequence.md
done
cd $R
timeout 600 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_bench_multirank.py -m gpu -q -p no:cacheprovider --timeout=200 > $OUT/pytest_sharded.log 2>&1; echo "exit $?" >> $OUT/pytest_sharded.log
tail -5 $OUT/pytest_sharded.log | cut -c1-300
