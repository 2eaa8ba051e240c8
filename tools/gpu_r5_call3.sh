#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05_c
timeout 600 python tools/w4_check.py > gpurun_out/r05_c/w4_check.txt 2>&1; echo "rc $?" >> gpurun_out/r05_c/w4_check.txt
tail -4 gpurun_out/r05_c/w4_check.txt
timeout 600 python tools/w4_diag.py > gpurun_out/r05_c/w4_diag.txt 2>&1; echo "rc $?" >> gpurun_out/r05_c/w4_diag.txt
cat gpurun_out/r05_c/w4_diag.txt
timeout 900 python -m pytest tests/test_gpu_kron.py -x -q -m gpu --timeout 300 2>&1 | tail -5
