#!/bin/bash
# experiment: stages walk the tensors last-first (what the predecessor wrote last is still in the last-level cache)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05_rev
for v in 0 1 3 7 15 8 0; do
  PSGDK_REV=$v timeout 600 python bench.py --steps 30 --warmup 10 --no-apply-only --no-cpu-baseline --no-secondary --no-peaks > gpurun_out/r05_rev/bench_rev_$v.json 2> gpurun_out/r05_rev/bench_rev_$v.err; echo "bench REV=$v rc $?"
  python - <<PY
import json
d = json.loads(open("gpurun_out/r05_rev/bench_rev_$v.json").read().strip().splitlines()[-1])
print("REV=$v ms_per_step", round(d["ms_per_step"],4), "median", round(d["ms_per_step_median"],4), "min", round(d["ms_per_step_min"],4), "gemm_ms", round(d["roofline"]["gemm_ms_per_step"],4))
print("   steps", [round(x,3) for x in d["config"]["step_device_ms"][:12]])
PY
done
