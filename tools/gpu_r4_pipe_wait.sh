#!/bin/bash
# round 4, calls 18 / 19 (HISTORICAL, product untouched; TIMING ONLY -- the variants compute garbage): what the waits, the DMA, the barriers, the fragment reads and the
# priority flips cost the 256 x 256 kernel's main loop, and the DMA / the per-K-step wait + barrier the 128 x 128 kernel's
R=$(pwd); OUT=$R/gpurun_out/r04_pipe_wait; mkdir -p $OUT; cd /tmp
for v in base v3 v4 v5 base; do
  lib=$R/psgd_torch_amd/libpsgdk_$v.so; [ $v = base ] && lib=$R/psgd_torch_amd/libpsgdk.so
  python $R/tools/gemm_parts_lib.py $lib 2>&1 | grep -v amdgpu.ids | tee -a $OUT/parts2.txt
done
for v in base k2 k3 base; do
  lib=$R/psgd_torch_amd/libpsgdk_$v.so; [ $v = base ] && lib=$R/psgd_torch_amd/libpsgdk.so
  python $R/tools/gemm_parts_lib.py $lib k128 2>&1 | grep -v amdgpu.ids | tee -a $OUT/parts2.txt
done
