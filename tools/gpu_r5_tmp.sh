#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05_k
timeout 1200 python -m pytest tests/test_gpu_kron.py tests/test_gpu_production_path.py tests/test_gpu_fullsize.py tests/test_gpu_eq.py -x -q -m gpu --timeout 600 2>&1 | tail -3
timeout 600 python tools/stage_bench.py small 0,56,7 2>&1 | tee gpurun_out/r05_k/stage_bench_small.txt | grep -v "^$"
timeout 600 python tools/stage_bench.py medium 0,56,7 2>&1 | tee gpurun_out/r05_k/stage_bench_medium.txt | grep -v "^$"
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-apply-only --no-cpu-baseline --no-secondary --no-peaks > gpurun_out/r05_k/bench_$i.json 2>/dev/null
python - <<PY
import json
d = json.loads(open("gpurun_out/r05_k/bench_$i.json").read().strip().splitlines()[-1])
print("   bench ms_per_step", round(d["ms_per_step"],4), "median", round(d["ms_per_step_median"],4), "min", round(d["ms_per_step_min"],4), "gemm_ms", round(d["roofline"]["gemm_ms_per_step"],4), "frac", round(d["roofline"]["frac"],4))
PY
done
