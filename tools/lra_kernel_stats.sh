#!/bin/bash
# round 6: per-kernel times of the LRA path (ViT-B, r = 10) in fp32 and bf16 on the current library
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r6u; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for dt in fp32 bf16; do
  fl=""; [ $dt = bf16 ] && fl="--bf16"
  rm -rf /tmp/p_lra_$dt
  rocprofv3 --kernel-trace --stats -d /tmp/p_lra_$dt -- python $R/bench.py --config vit-b-lra $fl --steps 6 --warmup 2 --no-peaks > $out/bench_$dt.json 2> $out/err_$dt
  python $R/tools/rocpd_stats.py $(find /tmp/p_lra_$dt -name "*.db" | head -1) > $out/stats_$dt.md
  echo "== $dt"; grep "lra_" $out/stats_$dt.md | cut -c1-150
done
