#!/bin/bash
# round 4, call 18 (HISTORICAL, product untouched; TIMING ONLY -- the variants compute garbage): what the phase-4 vmcnt(0) and the main loop's DMA cost the 256 x 256 kernel
R=$(pwd); OUT=$R/gpurun_out/r04_pipe_wait; mkdir -p $OUT; cd /tmp
for v in base v1 v2 base; do
  lib=$R/psgd_torch_amd/libpsgdk_$v.so; [ $v = base ] && lib=$R/psgd_torch_amd/libpsgdk.so
  python $R/tools/gemm_parts_lib.py $lib 2>&1 | grep -v amdgpu.ids | tee -a $OUT/parts.txt
done
