#!/bin/bash
# round 6, call A: the fused parameter update -- its own tests, the golden KWNS4 suites through it, and the A/B of the bench
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r6a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fused_update.py -x -q 2>&1 | tail -25 > $O/pytest_fused.log
timeout 900 python -m pytest tests/test_gpu_kron.py -x -q 2>&1 | tail -15 > $O/pytest_kron.log
for v in fuse nofuse fuse2 nofuse2; do
  f=""; case $v in nofuse*) f="--no-fuse";; esac
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-peaks $f > $O/bench_$v.json 2> $O/bench_$v.err
done
tail -3 $O/pytest_fused.log $O/pytest_kron.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6a/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, 'ms', round(d['ms_per_step'],4), 'median', round(d['ms_per_step_median'],4), 'apply_only', round(d['config']['apply_only_ms_per_step'],4), 'gemm_ms', d.get('roofline',{}).get('gemm_ms_per_step'), 'launches', d['config'].get('launches_per_step'))
    except Exception as e: print(f, 'ERR', e)
PY
