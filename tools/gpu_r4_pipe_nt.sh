#!/bin/bash
# round 4, call 17 (HISTORICAL, product untouched): the non-temporal bit on the 256 x 256 kernel's operand DMA (1 = the quad-dim operand, 2 = the lane-dim operand), built from a
# scratch copy of csrc/ as psgd_torch_amd/libpsgdk_p<mask>.so; GPT-2-small, A/B/.../A on one box.  Results: profiles/r04_i/README.md
OUT=$(pwd)/gpurun_out/r04_pipe_nt
R=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in base p1 p2 p3 base2; do
  lib=$R/psgd_torch_amd/libpsgdk_$v.so; [ $v = base ] && lib=$R/psgd_torch_amd/libpsgdk.so; [ $v = base2 ] && lib=$R/psgd_torch_amd/libpsgdk.so
  python $R/tools/bench_with_lib.py $lib --steps 30 --warmup 5 --no-cpu-baseline --no-peaks --no-apply-only 2> $OUT/bench_$v.err | tail -1 > $OUT/bench_$v.json
  rm -rf /tmp/p_$v
  rocprofv3 --kernel-trace --stats -d /tmp/p_$v -- python $R/tools/bench_with_lib.py $lib --steps 10 --warmup 3 --no-cpu-baseline --no-apply-only --no-peaks --no-roofline > /dev/null 2> $OUT/rocprof_$v.err
  python $R/tools/rocpd_sequence.py $(find /tmp/p_$v -name "*.db" | head -1) accumulate_kernel -3 > $OUT/step_sequence_$v.md
  python -c "
import json
d=json.loads(open('$OUT/bench_$v.json').read().strip().splitlines()[-1])
pipe=[l.split('|')[4].strip() for l in open('$OUT/step_sequence_$v.md') if 'gemm_nt_pipe' in l]
print('$v', 'median', round(d['ms_per_step_median'],4), 'min', round(d['ms_per_step_min'],4), 'pipe launches us', pipe)"
done
