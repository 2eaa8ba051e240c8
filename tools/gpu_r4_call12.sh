#!/bin/bash
# round 4, call 12: sharded / DTensor GPU tests after the load_state_dict change; per-pass kernel stats of the bf16 LRA step (lead 3 of DESIGN section 8)
OUT=$(pwd)/gpurun_out/r04_c12
R=$(pwd)
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_bench_multirank.py tests/test_gpu_dtensor.py tests/test_gpu_nlb.py -m gpu -q -p no:cacheprovider --timeout=200 > $OUT/pytest_multirank.log 2>&1; echo "exit $?" >> $OUT/pytest_multirank.log
tail -4 $OUT/pytest_multirank.log | cut -c1-300
cd /tmp
for dt in bf16 fp32; do
  rm -rf /tmp/p_lra_$dt
  flag=$( [ $dt = bf16 ] && echo --bf16 )
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_lra_$dt -- python $R/bench.py --config vit-b-lra $flag --steps 10 --warmup 3 --no-cpu-baseline --no-peaks > $OUT/bench_lra_${dt}_under_rocprof.json 2> $OUT/rocprof_lra_$dt.err
  python $R/tools/rocpd_stats.py $(find /tmp/p_lra_$dt -name "*.db" | head -1) > $OUT/vit-b-lra_${dt}_kernel_stats.md
  head -22 $OUT/vit-b-lra_${dt}_kernel_stats.md | cut -c1-160
done
