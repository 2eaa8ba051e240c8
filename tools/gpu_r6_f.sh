#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r6f; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/fuse_bench.py 2>&1 | tail -2 > $O/fuse_bench.txt
for v in fuse nofuse fuse2 nofuse2; do
  f=""; case $v in nofuse*) f="--no-fuse";; esac
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-peaks $f > $O/bench_$v.json 2> $O/bench_$v.err
done
cat $O/fuse_bench.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6f/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, 'ms', round(d['ms_per_step'],4), 'median', round(d['ms_per_step_median'],4), 'apply_only', round(d['config']['apply_only_ms_per_step'],4), 'gemm_ms', d.get('roofline',{}).get('gemm_ms_per_step'), 'launches', d['config'].get('launches_per_step'))
    except Exception as e: print(f, 'ERR', e)
PY
