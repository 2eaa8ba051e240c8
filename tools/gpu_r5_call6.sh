#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05_f
timeout 600 python tools/w4_check.py > gpurun_out/r05_f/w4_check.txt 2>&1; echo "rc $?" >> gpurun_out/r05_f/w4_check.txt
tail -3 gpurun_out/r05_f/w4_check.txt
timeout 600 python tools/stage_bench.py small 20,21,22,23,24,25,27 > gpurun_out/r05_f/stage_bench_small.txt 2>&1; echo "rc $?"
grep -i "upd_a\|app_a" gpurun_out/r05_f/stage_bench_small.txt
timeout 900 python -m pytest tests/test_gpu_kron.py tests/test_gpu_production_path.py tests/test_gpu_fullsize.py -x -q -m gpu --timeout 600 2>&1 | tail -4
for v in 1 0; do
  PSGDK_W4=$v timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_f/bench_w4_$v.json 2> gpurun_out/r05_f/bench_w4_$v.err; echo "bench W4=$v rc $?"
  python - <<PY
import json
d = json.loads(open("gpurun_out/r05_f/bench_w4_$v.json").read().strip().splitlines()[-1])
print("W4=$v ms_per_step", round(d["ms_per_step"],4), "median", round(d["ms_per_step_median"],4), "min", round(d["ms_per_step_min"],4), "gemm_ms", round(d["roofline"]["gemm_ms_per_step"],4), "frac", round(d["roofline"]["frac"],4))
PY
done
