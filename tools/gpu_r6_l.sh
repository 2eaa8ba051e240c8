#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r6l; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fused_update.py tests/test_gpu_kron.py tests/test_gpu_fullsize.py tests/test_gpu_production_path.py -q -x 2>&1 | tail -8 > $O/pytest.log
for v in a b c; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-peaks > $O/bench_$v.json 2> $O/bench_$v.err
done
timeout 600 python tools/fuse_bench.py 2>&1 | tail -2 > $O/fuse_bench.txt
tail -n 4 $O/pytest.log; cat $O/fuse_bench.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6l/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, 'ms', round(d['ms_per_step'],4), 'median', round(d['ms_per_step_median'],4), 'min', round(d['ms_per_step_min'],4), 'apply_only', round(d['config']['apply_only_ms_per_step'],4), 'gemm_ms', round(d['roofline']['gemm_ms_per_step'],4), d['roofline'].get('fused_update_launch',{}).get('ms_per_step'), d['roofline'].get('frac_excluding_fused_update_launch'))
    except Exception as e: print(f, 'ERR', e)
PY
