#!/bin/bash
# round 6: the packed two-rows-per-thread LRA passes (kernels_lra_pk.hiph) -- parity tests on the tree's library, then the ViT-B LRA
# config in bf16 on ab_libs/lib_base.so and the variants named on the command line (base first and last), then per-kernel times
tag=${1:-lrapk}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag; mkdir -p $out
cd $R
if [ -z "$NOTESTS" ]; then
timeout 1200 python -m pytest tests/test_gpu_lra.py tests/test_gpu_lra_sharded.py -x -q -m gpu > $out/pytest.log 2>&1; tail -5 $out/pytest.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "lra" > $out/pytest_full.log 2>&1; tail -3 $out/pytest_full.log
fi
# (NOBASE=1: libraries older than ABI 404 do not load any more -- the first variant named is then the reference, run first and last)
first=${NOBASE:+$1}; first=${first:-base}
for v in $first "$@" $first; do
  lib=$R/ab_libs/lib_$v.so
  for dt in bf16 fp32; do
    fl=""; [ $dt = bf16 ] && fl="--bf16"
    n=$(ls $out | grep -c "bench_${dt}_${v}_")
    python tools/bench_with_lib.py $lib --config vit-b-lra $fl --steps 32 --warmup 8 --no-cpu-baseline --no-peaks 2>> $out/bench.err | tail -1 > $out/bench_${dt}_${v}_$n.json
    python -c "
import json; d=json.load(open('$out/bench_${dt}_${v}_$n.json')); print('$v $dt', round(d['ms_per_step'],3), d['roofline']['frac'])"
  done
done
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/p_lra_$v
  rocprofv3 --kernel-trace --stats -d /tmp/p_lra_$v -- python $R/tools/bench_with_lib.py $R/ab_libs/lib_$v.so --config vit-b-lra --bf16 --steps 6 --warmup 2 --no-peaks --no-cpu-baseline > $out/prof_bench_$v.json 2> $out/prof_err_$v
  python $R/tools/rocpd_stats.py $(find /tmp/p_lra_$v -name "*.db" | head -1) > $out/stats_$v.md
  echo "== $v"; grep "lra" $out/stats_$v.md | cut -c1-160
done
