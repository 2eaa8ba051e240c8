#!/bin/bash
# round 4, final call: the whole GPU suite (a time limit per TEST), then the evidence set of profiles/r04_fin_*
OUT=gpurun_out/r04_fin
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1100 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=240 --durations=8 > $OUT/pytest_gpu_all.log 2>&1; echo "exit $?" >> $OUT/pytest_gpu_all.log
tail -25 $OUT/pytest_gpu_all.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 1500 bash tools/collect_profiles.sh r04_fin > $OUT/collect.log 2>&1; echo "collect exit $?"
cat $OUT/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('ms_per_step','ms_per_step_median','ms_per_step_min','kernel_ms_per_step','value')}, d['roofline']['frac'], d['roofline'].get('whole_step_frac_of_peak'), d.get('cpu_baseline',{}).get('value'))"
for c in gpt2-medium lenet5 gpt2-small-eq vit-b-lra; do python -c "import json;d=json.loads(open('$OUT/bench_$c.json').read().strip().splitlines()[-1]);print('$c', d['ms_per_step'], d.get('ms_per_step_median'), d['roofline']['frac'] if 'roofline' in d else None)"; done
python bench.py --config vit-b-lra --bf16 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_vit-b-lra_bf16.json 2>> $OUT/bench.err
tail -30 $OUT/step_sequence.md
