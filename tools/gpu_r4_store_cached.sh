#!/bin/bash
# round 4, call 21 (HISTORICAL, product untouched): cached instead of non-temporal output stores in the register-direct epilogue (2) and in the staged one too (3)
R=$(pwd); OUT=$R/gpurun_out/r04_pipe_wait; mkdir -p $OUT; cd /tmp
for v in base c2 c3 base; do
  lib=$R/psgd_torch_amd/libpsgdk_$v.so; [ $v = base ] && lib=$R/psgd_torch_amd/libpsgdk.so
  python $R/tools/bench_with_lib.py $lib --steps 30 --warmup 5 --no-cpu-baseline --no-peaks --no-apply-only --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('$v median', round(d['ms_per_step_median'],4), 'min', round(d['ms_per_step_min'],4), 'mean', round(d['ms_per_step'],4))" | tee -a $OUT/store_cached.txt
done
