#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05_e
timeout 600 python tools/stage_bench.py small 20,21,22,23,24,25,27 > gpurun_out/r05_e/stage_bench_small.txt 2>&1; echo "rc $?"
grep -i "upd_a\|app_a" gpurun_out/r05_e/stage_bench_small.txt
timeout 600 python tools/stage_bench.py medium 20,21,22,23,24,25,27 > gpurun_out/r05_e/stage_bench_medium.txt 2>&1; echo "rc $?"
grep -i "upd_a\|app_a" gpurun_out/r05_e/stage_bench_medium.txt
