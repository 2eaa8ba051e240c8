#!/bin/bash
# Round 3 final: the whole GPU suite on the final build, then the profile set.
OUT=gpurun_out/r03_final
mkdir -p $OUT
export TMPDIR=/tmp
timeout 420 python -m pytest tests -m gpu -q --durations=12 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "exit $?" >> $OUT/pytest_gpu.log
tail -20 $OUT/pytest_gpu.log
bash tools/collect_profiles.sh r03_zz > $OUT/collect.log 2>&1
head -c 400 gpurun_out/r03_zz/bench.json
