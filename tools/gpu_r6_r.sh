#!/bin/bash
# round 6: the dispatch sequence of an APPLY-ONLY step (preconditioner update gated off) of the GPT-2-small plan
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r6r; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_ao -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-peaks --no-secondary --no-roofline > $out/bench.json 2> $out/err
db=$(find /tmp/p_ao -name "*.db" | head -1)
python $R/tools/rocpd_sequence.py $db accumulate_kernel -3 > $out/apply_only_step_sequence.md
cat $out/apply_only_step_sequence.md
