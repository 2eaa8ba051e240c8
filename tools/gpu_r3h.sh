#!/bin/bash
# Round 3, GPU call H: 16-byte publish stores of the cooperative norm bound: route-vs-route soak, goldens, production path; A/B trace.
OUT=gpurun_out/r03h
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
timeout 240 python -m pytest tests/test_gpu_nlb.py tests/test_gpu_kron.py tests/test_gpu_eq.py -m gpu -q -p no:cacheprovider > $OUT/pytest_a.log 2>&1; echo "exit $?" >> $OUT/pytest_a.log
timeout 120 python -m pytest tests/test_gpu_production_path.py -m gpu -q -k "small_full_plan and True or fp32 or dumped" -p no:cacheprovider > $OUT/pytest_b.log 2>&1; echo "exit $?" >> $OUT/pytest_b.log
for v in 1 0; do
  ( cd /tmp && PSGDK_NLB_WIDE=$v rocprofv3 --kernel-trace --stats -d /tmp/p_w$v -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-apply-only --no-peaks > $R/$OUT/bench_wide$v.json 2>> $R/$OUT/rocprof.err
    db=$(find /tmp/p_w$v -name "*.db" | head -1); python $R/tools/rocpd_sequence.py $db accumulate_kernel -3 > $R/$OUT/step_sequence_wide$v.md )
done
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p_l5 -- python $R/bench.py --config lenet5 --steps 12 --warmup 4 --no-cpu-baseline --no-apply-only --no-peaks > /dev/null 2>> $R/$OUT/rocprof.err
  db=$(find /tmp/p_l5 -name "*.db" | head -1); python $R/tools/rocpd_sequence.py $db accumulate_kernel -3 > $R/$OUT/lenet5_step_sequence.md )
tail -4 $OUT/pytest_a.log; tail -4 $OUT/pytest_b.log; for v in 1 0; do echo wide=$v; grep nlb_coop $OUT/step_sequence_wide$v.md; tail -1 $OUT/step_sequence_wide$v.md; done; grep nlb_coop $OUT/lenet5_step_sequence.md
