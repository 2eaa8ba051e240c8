#!/bin/bash
TAG=${1:-r02c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/nlb_diag.py > $OUT/nlb_diag.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_nlb.py tests/test_gpu_dtensor.py -q -p no:cacheprovider > $OUT/pytest_fix.log 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/p_stats -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-apply-only > $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err
db=$(find /tmp/p_stats -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $db > $GRAFT_REPO_ROOT/$OUT/kernel_stats.md
python $GRAFT_REPO_ROOT/tools/rocpd_sequence.py $db accumulate_kernel -3 > $GRAFT_REPO_ROOT/$OUT/step_sequence.md
cd $GRAFT_REPO_ROOT
cat $OUT/nlb_diag.txt | tail -40; tail -5 $OUT/pytest_fix.log; cat $OUT/step_sequence.md
