#!/bin/bash
# Round 3, GPU call F: GPT-2-medium with every >= 768-tile stage on the 256 x 256 kernel (the transposed-space chain made Q' and RQ^T
# single-output problems) against the default tiling rule, same box.
TAG=${1:-r03f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
for v in default big; do
  if [ $v = big ]; then export PSGDK_BIG_MIN_TILES=768; else unset PSGDK_BIG_MIN_TILES; fi
  timeout 100 python bench.py --config gpt2-medium --steps 20 --warmup 5 --no-cpu-baseline --no-peaks 2>> $OUT/bench.err | tail -1 > $OUT/bench_medium_$v.json
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p_$v -- python $R/bench.py --config gpt2-medium --steps 8 --warmup 3 --no-cpu-baseline --no-apply-only --no-peaks > /dev/null 2>> $R/$OUT/rocprof.err
    db=$(find /tmp/p_$v -name "*.db" | head -1); python $R/tools/rocpd_sequence.py $db accumulate_kernel -3 > $R/$OUT/medium_step_sequence_$v.md )
done
unset PSGDK_BIG_MIN_TILES
timeout 100 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-peaks > $OUT/bench_small.json 2>> $OUT/bench.err
for v in default big; do head -c 230 $OUT/bench_medium_$v.json; echo; grep "gemm_nt\|rsub" $OUT/medium_step_sequence_$v.md; tail -1 $OUT/medium_step_sequence_$v.md; done
