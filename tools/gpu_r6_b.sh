#!/bin/bash
# round 6, call B: fused-update tests + stagger sweep of the fused launches
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r6b; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fused_update.py -x -q 2>&1 | tail -25 > $O/pytest_fused.log
for v in 0 1000 2000 3000 4000 0b nofuse; do
  f="--fuse-stagger ${v%b}"; case $v in nofuse) f="--no-fuse";; esac
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-peaks $f > $O/bench_$v.json 2> $O/bench_$v.err
done
tail -n 3 $O/pytest_fused.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6b/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, 'ms', round(d['ms_per_step'],4), 'median', round(d['ms_per_step_median'],4), 'apply_only', round(d['config']['apply_only_ms_per_step'],4), 'gemm_ms', d.get('roofline',{}).get('gemm_ms_per_step'), 'launches', d['config'].get('launches_per_step'))
    except Exception as e: print(f, 'ERR', e)
PY
