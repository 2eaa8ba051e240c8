#!/bin/bash
# round 4, call 4: the sharded path's GPU tests after the double-wait fix (row-split tensors, update after the apply, every balancing gate),
# the 256 x 128 tiling with its two-group main loop (v2), and where the 26-dim case spends its time on this host
OUT=gpurun_out/r04_call4
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
timeout 600 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_bench_multirank.py tests/test_gpu_c_abi.py -m gpu -q -p no:cacheprovider --timeout=180 > $OUT/pytest_sharded.log 2>&1; echo "exit $?" >> $OUT/pytest_sharded.log
tail -30 $OUT/pytest_sharded.log | cut -c1-300
timeout 200 python tools/gemm_mid_check.py > $OUT/mid_check.txt 2>&1; echo "exit $?" >> $OUT/mid_check.txt
grep -c "OK" $OUT/mid_check.txt; grep "FAIL" $OUT/mid_check.txt | head; tail -13 $OUT/mid_check.txt
timeout 200 python tools/stage_bench.py small 0,13,17,15 > $OUT/stage_bench_mid.txt 2>&1; echo "exit $?" >> $OUT/stage_bench_mid.txt
cat $OUT/stage_bench_mid.txt
for v in 0 1; do
  PSGDK_GEMM_MID=$v timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-apply-only --no-peaks > $OUT/bench_mid$v.json 2>> $OUT/bench.err
  echo "mid=$v"; python -c "import json;d=json.loads(open('$OUT/bench_mid$v.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_min'], d['roofline']['frac'])"
done
PSGDK_GEMM_MID=2 timeout 400 python -m pytest tests/test_gpu_kron.py tests/test_gpu_production_path.py -m gpu -q -p no:cacheprovider -x --timeout=200 > $OUT/pytest_mid_forced.log 2>&1; echo "exit $?" >> $OUT/pytest_mid_forced.log
tail -4 $OUT/pytest_mid_forced.log
PSGDK_SLOW_TESTS=1 timeout 170 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider -k "20_and_26 and 26" -o faulthandler_timeout=45 > $OUT/pytest_26dim.log 2>&1; echo "exit $?" >> $OUT/pytest_26dim.log
grep -n "File \"/root\|File \".*tests\|passed\|failed\|Timeout\|exit" $OUT/pytest_26dim.log | head -40
nproc; python -c "import torch; print(torch.get_num_threads(), torch.__config__.parallel_info()[:300])"
