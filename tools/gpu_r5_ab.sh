#!/bin/bash
# A/B of two builds of the library on the real plan: stage bench + bench.py (usage: gpu_r5_ab.sh libA.so libB.so)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05_ab
for rep in 1 2; do
for lib in "$@"; do
  cp "$lib" psgd_torch_amd/libpsgdk.so
  echo "== $lib"
  timeout 600 python tools/stage_bench.py small 0,7,8 2>&1 | grep -i "upd_a\|app_a\|gram\|qupd\|rq"
  timeout 600 python bench.py --steps 30 --warmup 10 --no-apply-only --no-cpu-baseline --no-secondary --no-peaks > gpurun_out/r05_ab/bench_$(basename $lib)_$rep.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open("gpurun_out/r05_ab/bench_$(basename $lib)_$rep.json").read().strip().splitlines()[-1])
print("   bench ms_per_step", round(d["ms_per_step"],4), "median", round(d["ms_per_step_median"],4), "min", round(d["ms_per_step_min"],4), "gemm_ms", round(d["roofline"]["gemm_ms_per_step"],4))
PY
done
done
