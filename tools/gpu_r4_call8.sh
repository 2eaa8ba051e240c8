#!/bin/bash
# round 4, call 8: the replica test with the gate pinned; where a sharded rank's step time goes (host enqueue vs chunks)
OUT=gpurun_out/r04_c8
mkdir -p $OUT
export TMPDIR=/tmp
for k in 1 2; do
  timeout 300 python -m pytest tests/test_gpu_dtensor.py tests/test_gpu_nlb.py -m gpu -q -p no:cacheprovider --timeout=120 -k "dtensor or reproducible" > $OUT/pytest_replicas_$k.log 2>&1; echo "exit $?" >> $OUT/pytest_replicas_$k.log
  tail -3 $OUT/pytest_replicas_$k.log | cut -c1-300
done
grep -h "differs by" $OUT/pytest_replicas_*.log | sort | uniq -c | sort -rn | head -12
R="python tools/rank_arithmetic.py"
timeout 300 $R --world 8 --chunks 1 --out $OUT/ra_w8_c1.json > $OUT/ra_w8_c1.md 2> $OUT/ra_w8_c1.err; tail -12 $OUT/ra_w8_c1.md
timeout 200 $R --world 8 --chunks 2 --ranks 0,4 > $OUT/ra_w8_c2.md 2> $OUT/ra_w8_c2.err; tail -5 $OUT/ra_w8_c2.md
timeout 200 $R --world 8 --ranks 4 --cprofile > $OUT/ra_w8_c4_prof.md 2> $OUT/ra_w8_c4_prof.err; tail -4 $OUT/ra_w8_c4_prof.md; head -60 $OUT/ra_w8_c4_prof.err | cut -c1-200
timeout 200 $R --world 8 --chunks 1 --ranks 4 --cprofile > $OUT/ra_w8_c1_prof.md 2> $OUT/ra_w8_c1_prof.err; head -45 $OUT/ra_w8_c1_prof.err | cut -c1-200
timeout 200 $R --world 2 --chunks 1 > $OUT/ra_w2_c1.md 2> $OUT/ra_w2_c1.err; tail -5 $OUT/ra_w2_c1.md
