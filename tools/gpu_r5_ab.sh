#!/bin/bash
# bench A/B: PSGDK_W4=1 / 0, twice each
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05_ab
for v in 1 0 1 0; do
  PSGDK_W4=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-apply-only > gpurun_out/r05_ab/bench_w4_$v.json 2> gpurun_out/r05_ab/bench_w4_$v.err; echo "bench W4=$v rc $?"
  python - <<PY
import json
d = json.loads(open("gpurun_out/r05_ab/bench_w4_$v.json").read().strip().splitlines()[-1])
print("W4=$v ms_per_step", round(d["ms_per_step"],4), "median", round(d["ms_per_step_median"],4), "min", round(d["ms_per_step_min"],4), "gemm_ms", round(d["roofline"]["gemm_ms_per_step"],4), "frac", round(d["roofline"]["frac"],4))
PY
done
