#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05_b
timeout 600 python tools/w4_diag.py > gpurun_out/r05_b/w4_diag.txt 2>&1; echo "rc $?" >> gpurun_out/r05_b/w4_diag.txt
cat gpurun_out/r05_b/w4_diag.txt
