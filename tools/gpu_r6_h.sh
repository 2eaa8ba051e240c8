#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r6h; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lra.py tests/test_gpu_lra_sharded.py -q -x 2>&1 | tail -30 > $O/pytest_lra.log
timeout 300 python bench.py --config vit-b-lra --steps 10 --warmup 5 --no-peaks > $O/bench_lra_fp32.json 2> $O/err1
timeout 300 python bench.py --config vit-b-lra --bf16 --steps 10 --warmup 5 --no-peaks > $O/bench_lra_bf16.json 2> $O/err2
tail -n 12 $O/pytest_lra.log
python - <<'PY'
import json
for f in ('fp32','bf16'):
    try:
        d=json.loads([l for l in open(f'gpurun_out/r6h/bench_lra_{f}.json') if l.startswith('{')][-1])
        print(f, 'ms', d['ms_per_step'], d['roofline'])
    except Exception as e: print(f,'ERR',e)
PY
