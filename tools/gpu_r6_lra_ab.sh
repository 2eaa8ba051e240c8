#!/bin/bash
# round 6 A/B of the LRA path inside one call: ab_libs/lib_base.so (HEAD) against the working tree's library; fp32 and bf16, base first and last
tag=${1:-lra_ab}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag; mkdir -p $out
cd $R
if [ $# -gt 0 ]; then timeout 1500 python -m pytest "$@" -x -q -m gpu > $out/pytest.log 2>&1; tail -3 $out/pytest.log; fi
for v in base new base new; do
  lib=$R/psgd_torch_amd/libpsgdk.so; [ $v = base ] && lib=$R/ab_libs/lib_base.so
  for dt in fp32 bf16; do
    fl=""; [ $dt = bf16 ] && fl="--bf16"
    n=$(ls $out | grep -c "bench_${dt}_${v}")
    python tools/bench_with_lib.py $lib --config vit-b-lra $fl --steps 32 --warmup 8 --no-cpu-baseline --no-peaks 2>> $out/bench.err | tail -1 > $out/bench_${dt}_${v}_$n.json
    python -c "
import json; d=json.load(open('$out/bench_${dt}_${v}_$n.json')); print('$v $dt', round(d['ms_per_step'],3), d['roofline']['frac'])"
  done
done
