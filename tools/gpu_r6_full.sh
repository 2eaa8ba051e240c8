#!/bin/bash
# the whole -m gpu suite + the driver's bench command
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r6full; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/pytest_gpu_all.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench.err
tail -n 5 $O/pytest_gpu_all.log
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r6full/bench_driver_command.json') if l.startswith('{')][-1])
print('ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'min', d['ms_per_step_min'], 'apply_only', d['config']['apply_only_ms_per_step'])
print('roofline', {k:d['roofline'][k] for k in ('achieved','frac','gemm_ms_per_step','whole_step_frac_of_peak','launches_per_step')})
print('secondary', json.dumps(d['config'].get('secondary'))[:1500])
PY
