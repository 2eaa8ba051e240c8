#!/bin/bash
# round 6: every stage on the 128 x 128 tiling (PSGDK_BIG_MIN_TILES huge) against the bound tilings -- step and apply-only sequences
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r6s; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in bound small; do
  [ $v = small ] && export PSGDK_BIG_MIN_TILES=1000000000
  rm -rf /tmp/p_$v
  rocprofv3 --kernel-trace --stats -d /tmp/p_$v -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-peaks --no-secondary --no-roofline > $out/bench_$v.json 2> $out/err_$v
  db=$(find /tmp/p_$v -name "*.db" | head -1)
  python $R/tools/rocpd_sequence.py $db accumulate_kernel -3 > $out/apply_only_sequence_$v.md
  python $R/tools/rocpd_sequence.py $db accumulate_kernel 5 > $out/step_sequence_$v.md
  echo "== $v"; cat $out/apply_only_sequence_$v.md | cut -c1-100; grep -n "gemm_nt\|kernel time" $out/step_sequence_$v.md | cut -c1-110
done
