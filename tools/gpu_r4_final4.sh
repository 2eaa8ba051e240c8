#!/bin/bash
# round 4, the last call: the library with the non-temporal hints -- whole GPU suite (a time limit per TEST), smoke, the evidence set (bench lines, rocprofv3
# stats, PMC traffic stamped with this library's hash)
OUT=gpurun_out/r04_fin4
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1100 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=240 --durations=5 > $OUT/pytest_gpu_all.log 2>&1; echo "exit $?" >> $OUT/pytest_gpu_all.log
tail -10 $OUT/pytest_gpu_all.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 900 bash tools/collect_profiles.sh r04_fin4 > $OUT/collect.log 2>&1; echo "collect exit $?"
cp $OUT/pmc_traffic.json profiles/pmc_traffic_latest.json; cp $OUT/pmc_traffic_vit-b-lra.json profiles/pmc_traffic_vit-b-lra_latest.json
python bench.py 2> $OUT/bench_final.err | tail -1 > $OUT/bench_final.json
for f in bench bench_final bench_no_events; do python -c "import json;d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]);print('$f', round(d['ms_per_step'],4), round(d['ms_per_step_median'],4), round(d['ms_per_step_min'],4), d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('traffic_stale'), d.get('roofline',{}).get('traffic_library_sha256'))"; done
for c in gpt2-medium lenet5 gpt2-small-eq vit-b-lra; do python -c "import json;d=json.loads(open('$OUT/bench_$c.json').read().strip().splitlines()[-1]);print('$c', d['ms_per_step'], d.get('ms_per_step_median'), d['roofline']['frac'] if 'roofline' in d else None)"; done
tail -25 $OUT/step_sequence.md
