#!/bin/bash
# full GPU suite + the driver's bench command
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05_full
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 600 > gpurun_out/r05_full/pytest_gpu.log 2>&1; echo "pytest rc $?"
tail -4 gpurun_out/r05_full/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_full/bench.json 2> gpurun_out/r05_full/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_full/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", round(d["ms_per_step"],4), "median", round(d["ms_per_step_median"],4), "min", round(d["ms_per_step_min"],4), "value", round(d["value"],2))
print("roofline", {k: d["roofline"][k] for k in ("frac", "gemm_ms_per_step", "steps_with_events", "whole_step_frac_of_peak")})
print("steps", d["config"]["step_device_ms"])
print("secondary", json.dumps(d["config"].get("secondary"), indent=1)[:1500])
PY
