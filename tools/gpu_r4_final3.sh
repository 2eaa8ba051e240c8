#!/bin/bash
# round 4, last call: the whole GPU suite on the final tree (a time limit per TEST), smoke, the bench line with the PMC traffic of THIS library
# (profiles/pmc_traffic_latest.json was collected on it in call 7), three repeats for the spread, and the per-rank table at the new default
OUT=gpurun_out/r04_fin3
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1100 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=240 --durations=5 > $OUT/pytest_gpu_all.log 2>&1; echo "exit $?" >> $OUT/pytest_gpu_all.log
tail -10 $OUT/pytest_gpu_all.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
python bench.py 2> $OUT/bench.err | tail -1 > $OUT/bench.json
for k in 2 3 4; do python bench.py --no-cpu-baseline 2>> $OUT/bench.err | tail -1 > $OUT/bench_repeat$k.json; done
for f in bench bench_repeat2 bench_repeat3 bench_repeat4; do python -c "import json;d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]);print('$f', round(d['ms_per_step'],4), round(d['ms_per_step_median'],4), round(d['ms_per_step_min'],4), round(d['roofline']['frac'],4), d['roofline']['traffic_stale'], d['roofline']['traffic_library_sha256'])"; done
timeout 400 python tools/rank_arithmetic.py --world 8 --out $OUT/rank_arithmetic_w8_default.json > $OUT/rank_arithmetic_w8_default.md 2> $OUT/rank_arithmetic_w8_default.err; cat $OUT/rank_arithmetic_w8_default.md
