#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05_h
timeout 600 python tools/w4_check.py > gpurun_out/r05_h/w4_check.txt 2>&1; echo "rc $?" >> gpurun_out/r05_h/w4_check.txt
tail -2 gpurun_out/r05_h/w4_check.txt
PSGDK_W4=1 timeout 600 python tools/stage_bench.py small 0,5,6,7,8 > gpurun_out/r05_h/stage_bench_w4.txt 2>&1; echo "rc $?"
grep -i "upd_a\|app_a" gpurun_out/r05_h/stage_bench_w4.txt
PSGDK_W4=0 timeout 600 python tools/stage_bench.py small 0,7,8 > gpurun_out/r05_h/stage_bench_pipe.txt 2>&1; echo "rc $?"
grep -i "upd_a\|app_a" gpurun_out/r05_h/stage_bench_pipe.txt
W4_DIAG_FIRST=1 timeout 600 python tools/w4_diag.py 2>&1 | head -20
