#!/bin/bash
# per-kernel times of the LRA update + apply at the wider rank classes
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for r in 32 64; do
  rocprofv3 --kernel-trace --stats -d /tmp/p_lra$r -- python $R/bench.py --config vit-b-lra --lra-rank $r --steps 3 --warmup 1 > /dev/null 2>&1
  python $R/tools/rocpd_stats.py $(find /tmp/p_lra$r -name "*.db" | head -1) | grep "lra_\|total kernel" | cut -c1-140 > $R/gpurun_out/lra_r${r}_kernel_stats.md
  echo "rank $r"; cat $R/gpurun_out/lra_r${r}_kernel_stats.md
done
