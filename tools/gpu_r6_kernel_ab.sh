#!/bin/bash
# round 6: per-kernel A/B of library builds on the headline config under rocprofv3 (kernel averages over 12 traced steps), interleaved:
#   tools/gpu_r6_kernel_ab.sh <tag> <reps> <kernel substring> <variant> ...
tag=$1; reps=$2; pat=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 $reps); do
  for v in "$@"; do
    rm -rf /tmp/p_$v
    rocprofv3 --kernel-trace --stats -d /tmp/p_$v -- python $R/tools/bench_with_lib.py $R/ab_libs/lib_$v.so --steps 12 --warmup 4 --no-cpu-baseline --no-apply-only --no-peaks --no-secondary > $out/bench_${v}_$i.json 2> $out/err_${v}_$i
    python $R/tools/rocpd_stats.py $(find /tmp/p_$v -name "*.db" | head -1) > $out/stats_${v}_$i.md
    echo "$v $i $(python -c "
import json; d=json.loads([l for l in open('$out/bench_${v}_$i.json') if l.startswith('{')][-1]); print(round(d['ms_per_step'],4), round(d['ms_per_step_median'],4))") $(grep "$pat" $out/stats_${v}_$i.md | cut -d'|' -f3-6 | tr '\n' ';')"
  done
done
