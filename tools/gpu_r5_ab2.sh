#!/bin/bash
cd "$GRAFT_REPO_ROOT"
cp psgd_torch_amd/libpsgdk_new.so psgd_torch_amd/libpsgdk.so
timeout 900 python -m pytest tests/test_gpu_kron.py tests/test_gpu_production_path.py -x -q -m gpu --timeout 600 2>&1 | tail -3
timeout 300 python tools/w4_check.py 2>&1 | tail -2
bash tools/gpu_r5_ab.sh psgd_torch_amd/libpsgdk_base.so psgd_torch_amd/libpsgdk_new.so
cp psgd_torch_amd/libpsgdk_new.so psgd_torch_amd/libpsgdk.so
