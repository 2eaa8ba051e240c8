#!/bin/bash
# round 4, call 2: the 256 x 128 tiling (gemm_nt_mid_kernel): correctness, per-stage A/B inside the bound GPT-2-small plan, the step with and
# without it; then the whole GPU suite on the new defaults (K-split small launches, early-vector LRA reductions, variants removed)
OUT=gpurun_out/r04_call2
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
timeout 300 python tools/gemm_mid_check.py > $OUT/mid_check.txt 2>&1; echo "exit $?" >> $OUT/mid_check.txt
grep -c "OK" $OUT/mid_check.txt; grep "FAIL" $OUT/mid_check.txt | head -20; tail -14 $OUT/mid_check.txt
timeout 300 python tools/stage_bench.py small 0,13,17,15,16,8 > $OUT/stage_bench_mid.txt 2>&1; echo "exit $?" >> $OUT/stage_bench_mid.txt
cat $OUT/stage_bench_mid.txt
for v in 0 1; do
  PSGDK_GEMM_MID=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-apply-only --no-peaks > $OUT/bench_mid$v.json 2>> $OUT/bench.err
  ( cd /tmp && PSGDK_GEMM_MID=$v timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_m$v -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-apply-only --no-peaks > /dev/null 2>> $R/$OUT/rocprof.err
    db=$(find /tmp/p_m$v -name "*.db" | head -1); python $R/tools/rocpd_sequence.py $db accumulate_kernel -3 > $R/$OUT/step_sequence_mid$v.md )
  echo "mid=$v"; python -c "import json;d=json.loads(open('$OUT/bench_mid$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['roofline']['frac'])"
  cat $OUT/step_sequence_mid$v.md | tail -30
done
# parity: every grouped-GEMM stage that can takes the middle tiling (threshold 2 tiles)
PSGDK_GEMM_MID=2 timeout 600 python -m pytest tests/test_gpu_kron.py tests/test_gpu_eq.py tests/test_gpu_production_path.py -m gpu -q -p no:cacheprovider -x > $OUT/pytest_mid_forced.log 2>&1; echo "exit $?" >> $OUT/pytest_mid_forced.log
tail -5 $OUT/pytest_mid_forced.log
PSGDK_GEMM_MID=1 timeout 400 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -x -k "gpt2_small or medium or block" > $OUT/pytest_mid_fullsize.log 2>&1; echo "exit $?" >> $OUT/pytest_mid_fullsize.log
tail -5 $OUT/pytest_mid_fullsize.log
# the whole suite on the defaults
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu_all.log 2>&1; echo "exit $?" >> $OUT/pytest_gpu_all.log
tail -8 $OUT/pytest_gpu_all.log
