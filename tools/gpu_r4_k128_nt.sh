#!/bin/bash
# round 4, call 20 (HISTORICAL, product untouched): the non-temporal bit on the 128 x 128 kernel's operand DMA (1 = A, 2 = B, 3 = both), built from a scratch copy
R=$(pwd); OUT=$R/gpurun_out/r04_pipe_wait; mkdir -p $OUT; cd /tmp
for v in base n1 n2 n3 base; do
  lib=$R/psgd_torch_amd/libpsgdk_$v.so; [ $v = base ] && lib=$R/psgd_torch_amd/libpsgdk.so
  python $R/tools/gemm_parts_lib.py $lib k128 2>&1 | grep -v amdgpu.ids | tee -a $OUT/parts3.txt
done
