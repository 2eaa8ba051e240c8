#!/bin/bash
# EQ geometry round: parity, then the GPT-2-small EQ bench with the bf16 solve and (A/B) the fp32-core solve, kernel stats
tag=${1:-eq}
out=gpurun_out/$tag; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_eq.py -x -q > $out/pytest_eq.log 2>&1; echo "exit $?" >> $out/pytest_eq.log
python bench.py --config gpt2-small-eq --steps 20 --warmup 5 > $out/bench_eq.json 2> $out/bench_eq.err
PSGDK_TRSM=f32 python bench.py --config gpt2-small-eq --steps 20 --warmup 5 > $out/bench_eq_f32cores.json 2> $out/bench_eq_f32cores.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof -- python $GRAFT_REPO_ROOT/bench.py --config gpt2-small-eq --steps 8 --warmup 2 > $GRAFT_REPO_ROOT/$out/prof.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find $out/prof -name "*_results.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" > $out/eq_kernel_stats.md 2>> $out/prof.log && python tools/rocpd_sequence.py "$db" > $out/eq_step_sequence.md 2>> $out/prof.log
rm -rf $out/prof
tail -n 4 $out/pytest_eq.log
for f in $out/bench_eq.json $out/bench_eq_f32cores.json; do python -c "
import json,sys
z=json.load(open('$f')); print('$f', z['ms_per_step'], z['value'])"; done
head -25 $out/eq_kernel_stats.md
