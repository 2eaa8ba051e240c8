#!/bin/bash
# round 5, call 4: full GPU suite on the w4 + staged-store build, bench A/B (PSGDK_W4=0 / 1)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05_d
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 600 > gpurun_out/r05_d/pytest_gpu.log 2>&1; echo "pytest rc $?"
tail -5 gpurun_out/r05_d/pytest_gpu.log
for v in 1 0 1 0; do
  PSGDK_W4=$v timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_d/bench_w4_$v.json 2> gpurun_out/r05_d/bench_w4_$v.err; echo "bench W4=$v rc $?"
  python - <<PY
import json
d = json.loads(open("gpurun_out/r05_d/bench_w4_$v.json").read().strip().splitlines()[-1])
print("W4=$v ms_per_step", round(d["ms_per_step"],4), "median", round(d["ms_per_step_median"],4), "min", round(d["ms_per_step_min"],4), "gemm_ms", round(d["roofline"]["gemm_ms_per_step"],4), "frac", round(d["roofline"]["frac"],4))
print("steps", d["config"]["step_device_ms"])
PY
done
