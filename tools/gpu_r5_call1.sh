#!/bin/bash
# round 5, call 1: the four-wave 256 x 256 kernel (correctness vs the 128 x 128 tiling, timing of its variants) + the bench's per-step timeline
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05_a
export TMPDIR=/tmp
timeout 600 python tools/w4_check.py > gpurun_out/r05_a/w4_check.txt 2>&1; echo "w4_check rc $?" >> gpurun_out/r05_a/w4_check.txt
tail -25 gpurun_out/r05_a/w4_check.txt
timeout 600 python tools/w4_bench.py > gpurun_out/r05_a/w4_bench.txt 2>&1; echo "w4_bench rc $?" >> gpurun_out/r05_a/w4_bench.txt
cat gpurun_out/r05_a/w4_bench.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_a/bench.json 2> gpurun_out/r05_a/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_a/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "median", d["ms_per_step_median"], "min", d["ms_per_step_min"])
print("steps", d["config"]["step_device_ms"])
PY
