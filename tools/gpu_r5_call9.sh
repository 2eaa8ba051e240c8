#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05_i
timeout 300 python tools/w4p_check.py > gpurun_out/r05_i/w4p_check.txt 2>&1; echo "rc $?" >> gpurun_out/r05_i/w4p_check.txt
tail -30 gpurun_out/r05_i/w4p_check.txt
timeout 300 python tools/w4_diag.py > gpurun_out/r05_i/w4_diag.txt 2>&1; echo "rc $?" >> gpurun_out/r05_i/w4_diag.txt
cat gpurun_out/r05_i/w4_diag.txt
