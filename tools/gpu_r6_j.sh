#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r6j; mkdir -p $O
export TMPDIR=/tmp
for v in base 1052576 1050576 base2 1052576b; do
  f=""; case $v in base*) f="";; *) f="--fuse-stagger ${v%b}";; esac
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-peaks $f > $O/bench_$v.json 2> $O/bench_$v.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6j/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, 'ms', round(d['ms_per_step'],4), 'median', round(d['ms_per_step_median'],4), 'min', round(d['ms_per_step_min'],4), 'apply_only', round(d['config']['apply_only_ms_per_step'],4), 'gemm_ms', round(d['roofline']['gemm_ms_per_step'],4), d['roofline'].get('fused_update_launch',{}).get('ms_per_step'), d['roofline'].get('frac_excluding_fused_update_launch'))
    except Exception as e: print(f, 'ERR', e)
PY
