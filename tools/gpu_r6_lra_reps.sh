#!/bin/bash
# round 6: repeated, interleaved A/B of library builds on the ViT-B LRA config: DT=bf16|fp32|both tools/gpu_r6_lra_reps.sh <tag> <reps> <variant> ...
tag=$1; reps=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag; mkdir -p $out
cd $R
python tools/bench_with_lib.py $R/ab_libs/lib_$1.so --config vit-b-lra --bf16 --steps 16 --warmup 8 --no-cpu-baseline --no-peaks > /dev/null 2>&1   # (warms the box)
DT=${DT:-bf16}; [ $DT = both ] && DT="bf16 fp32"
for i in $(seq 1 $reps); do
  for v in "$@"; do
    for dt in $DT; do
      fl=""; [ $dt = bf16 ] && fl="--bf16"
      python tools/bench_with_lib.py $R/ab_libs/lib_$v.so --config vit-b-lra $fl --steps 32 --warmup 8 --no-cpu-baseline --no-peaks 2>> $out/bench.err | tail -1 > $out/bench_${dt}_${v}_$i.json
      python -c "
import json; d=json.load(open('$out/bench_${dt}_${v}_$i.json')); print('$v', '$dt', $i, round(d['ms_per_step'],3))"
    done
  done
done
