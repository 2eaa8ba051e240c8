#!/bin/bash
# round 4, call 7 (after the norm bound's row sums left the LDS atomics): the replica tests first, the whole GPU suite (a time limit per
# TEST), the evidence set of profiles/r04_fin_*, and one rank's arithmetic of a 2 / 4 / 8-rank job (tools/rank_arithmetic.py)
OUT=gpurun_out/r04_fin2
mkdir -p $OUT
export TMPDIR=/tmp
for k in 1 2 3; do
  timeout 300 python -m pytest tests/test_gpu_dtensor.py tests/test_gpu_nlb.py -m gpu -q -p no:cacheprovider --timeout=120 -k "dtensor or reproducible" > $OUT/pytest_replicas_$k.log 2>&1; echo "exit $?" >> $OUT/pytest_replicas_$k.log
  tail -4 $OUT/pytest_replicas_$k.log | cut -c1-400
done
grep -h "differs by" $OUT/pytest_replicas_*.log | sort | uniq -c | sort -rn | head -20
timeout 1100 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=240 --durations=8 > $OUT/pytest_gpu_all.log 2>&1; echo "exit $?" >> $OUT/pytest_gpu_all.log
tail -14 $OUT/pytest_gpu_all.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 1500 bash tools/collect_profiles.sh r04_fin2 > $OUT/collect.log 2>&1; echo "collect exit $?"
cat $OUT/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('ms_per_step','ms_per_step_median','ms_per_step_min','kernel_ms_per_step','value')}, d['roofline']['frac'], d['roofline'].get('whole_step_frac_of_peak'), d.get('cpu_baseline',{}).get('value'))"
for c in gpt2-medium lenet5 gpt2-small-eq vit-b-lra; do python -c "import json;d=json.loads(open('$OUT/bench_$c.json').read().strip().splitlines()[-1]);print('$c', d['ms_per_step'], d.get('ms_per_step_median'), d['roofline']['frac'] if 'roofline' in d else None)"; done
python bench.py --config vit-b-lra --bf16 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_vit-b-lra_bf16.json 2>> $OUT/bench.err
grep -n nlb_coop $OUT/step_sequence.md
for w in 8 4 2; do
  timeout 400 python tools/rank_arithmetic.py --world $w --out $OUT/rank_arithmetic_w$w.json > $OUT/rank_arithmetic_w$w.md 2> $OUT/rank_arithmetic_w$w.err; echo "rank_arithmetic $w exit $?"
  tail -3 $OUT/rank_arithmetic_w$w.md
done
timeout 300 python tools/rank_arithmetic.py --world 8 --no-row-split --out $OUT/rank_arithmetic_w8_nosplit.json > $OUT/rank_arithmetic_w8_nosplit.md 2> $OUT/rank_arithmetic_w8_nosplit.err
tail -3 $OUT/rank_arithmetic_w8_nosplit.md
