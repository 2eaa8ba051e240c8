#!/bin/bash
# GPT-2-medium: per-stage kernel times with the default tiling choice vs every stage on the 128 x 128 kernel
out=gpurun_out/${1:-medab}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in default all128; do
  if [ $v = all128 ]; then export PSGDK_BIG_MIN_TILES=100000000; else unset PSGDK_BIG_MIN_TILES; fi
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof_$v -- python $GRAFT_REPO_ROOT/bench.py --config gpt2-medium --steps 6 --warmup 2 --no-cpu-baseline --no-apply-only > $GRAFT_REPO_ROOT/$out/bench_$v.json 2> $GRAFT_REPO_ROOT/$out/bench_$v.err
  db=$(find $GRAFT_REPO_ROOT/$out/prof_$v -name "*_results.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/rocpd_sequence.py "$db" > $GRAFT_REPO_ROOT/$out/seq_$v.md 2>> $GRAFT_REPO_ROOT/$out/bench_$v.err
  rm -rf $GRAFT_REPO_ROOT/$out/prof_$v
done
cd $GRAFT_REPO_ROOT
paste <(cut -d'|' -f3,4,5 $out/seq_default.md | cut -c1-60) <(cut -d'|' -f4,5 $out/seq_all128.md) | head -45
