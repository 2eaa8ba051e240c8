#!/bin/bash
# Round 3, GPU call A: the new parity tests first (production noise path, dQ switch, strided params, LRA clip-last, chain A/B),
# bench lines with the transposed-space chain and with the legacy one, a kernel trace of each, then the rest of the suite.
TAG=${1:-r03a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
python -c "import torch; print(torch.__version__, torch.cuda.get_device_name(0), torch.cuda.device_count())" > $OUT/env.txt 2>&1
nproc >> $OUT/env.txt; grep -m1 "model name" /proc/cpuinfo >> $OUT/env.txt; free -g >> $OUT/env.txt
timeout 900 python -m pytest tests/test_gpu_production_path.py -q -x --durations=10 -p no:cacheprovider > $OUT/pytest_prod.log 2>&1
echo "exit $?" >> $OUT/pytest_prod.log
timeout 300 python bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
PSGDK_CHAIN=legacy timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-peaks > $OUT/bench_legacy_chain.json 2>> $OUT/bench.err
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p_new -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-apply-only --no-peaks > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/rocprof.err
  db=$(find /tmp/p_new -name "*.db" | head -1); python $R/tools/rocpd_stats.py $db > $R/$OUT/kernel_stats.md; python $R/tools/rocpd_sequence.py $db accumulate_kernel -3 > $R/$OUT/step_sequence.md )
( cd /tmp && PSGDK_CHAIN=legacy rocprofv3 --kernel-trace --stats -d /tmp/p_old -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-apply-only --no-peaks > /dev/null 2>> $R/$OUT/rocprof.err
  db=$(find /tmp/p_old -name "*.db" | head -1); python $R/tools/rocpd_sequence.py $db accumulate_kernel -3 > $R/$OUT/step_sequence_legacy_chain.md )
timeout 1500 python -m pytest tests -m gpu -q --durations=15 -p no:cacheprovider --deselect tests/test_gpu_production_path.py > $OUT/pytest_rest.log 2>&1
echo "exit $?" >> $OUT/pytest_rest.log
tail -5 $OUT/pytest_prod.log; tail -5 $OUT/pytest_rest.log; head -c 1200 $OUT/bench.json; echo; head -c 400 $OUT/bench_legacy_chain.json
