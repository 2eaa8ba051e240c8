#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05_g
PSGDK_W4=1 timeout 600 python tools/stage_bench.py small 0,5,6,7,8 > gpurun_out/r05_g/stage_bench_w4.txt 2>&1; echo "rc $?"
grep -i "upd_a\|app_a" gpurun_out/r05_g/stage_bench_w4.txt
PSGDK_W4=0 timeout 600 python tools/stage_bench.py small 0,5,6,7,8 > gpurun_out/r05_g/stage_bench_pipe.txt 2>&1; echo "rc $?"
grep -i "upd_a\|app_a" gpurun_out/r05_g/stage_bench_pipe.txt
