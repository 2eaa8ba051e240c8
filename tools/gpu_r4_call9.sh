#!/bin/bash
# round 4, call 9: per-rank step time against the chunk count, GPT-2-medium (and GPT-2-small at 4 ranks): calibrates the default chunk rule
OUT=gpurun_out/r04_c9
mkdir -p $OUT
export TMPDIR=/tmp
R="python tools/rank_arithmetic.py --steps 20"
for c in 1 2 4; do
  timeout 300 $R --config gpt2-medium --world 8 --chunks $c --ranks 0,4 > $OUT/ra_med_w8_c$c.md 2> $OUT/ra_med_w8_c$c.err; echo "medium world 8 chunks $c"; tail -4 $OUT/ra_med_w8_c$c.md
  timeout 300 $R --config gpt2-medium --world 2 --chunks $c --ranks 0 > $OUT/ra_med_w2_c$c.md 2> $OUT/ra_med_w2_c$c.err; echo "medium world 2 chunks $c"; tail -3 $OUT/ra_med_w2_c$c.md
done
for c in 1 2; do
  timeout 300 $R --world 4 --chunks $c --ranks 0,2 > $OUT/ra_w4_c$c.md 2> $OUT/ra_w4_c$c.err; echo "small world 4 chunks $c"; tail -4 $OUT/ra_w4_c$c.md
done
