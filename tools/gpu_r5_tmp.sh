#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05_l
cp psgd_torch_amd/libpsgdk_base.so psgd_torch_amd/libpsgdk.so
echo "== base"; timeout 300 python tools/ew_check.py 2>&1 | tail -1
cp psgd_torch_amd/libpsgdk_new.so psgd_torch_amd/libpsgdk.so
echo "== new"; timeout 300 python tools/ew_check.py 2>&1 | tail -1
echo "== new, general path"; PSGDK_EW_DBG=2 timeout 300 python tools/ew_check.py 2>&1 | tail -1
echo "== new fp32"; timeout 300 python tools/ew_check.py fp32 2>&1 | tail -1
cp psgd_torch_amd/libpsgdk_base.so psgd_torch_amd/libpsgdk.so
echo "== base fp32"; timeout 300 python tools/ew_check.py fp32 2>&1 | tail -1
cp psgd_torch_amd/libpsgdk_new.so psgd_torch_amd/libpsgdk.so
timeout 300 python tools/ew_bench.py 2>&1 | tail -3
echo "== new, general path"; PSGDK_EW_DBG=2 timeout 300 python tools/ew_bench.py 2>&1 | tail -3
