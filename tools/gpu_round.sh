#!/bin/bash
# One gpurun call: the new parity tests first, then the bench line, then the rest of the GPU suite.  Everything the builder wants
# to read back goes under gpurun_out/$TAG/.
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import torch; print(torch.__version__, torch.cuda.get_device_name(0))" > $OUT/env.txt 2>&1
nproc >> $OUT/env.txt; grep -m1 "model name" /proc/cpuinfo >> $OUT/env.txt; free -g >> $OUT/env.txt
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_nlb.py tests/test_gpu_dtensor.py -q --durations=20 -p no:cacheprovider > $OUT/pytest_new.log 2>&1
echo "exit $?" >> $OUT/pytest_new.log
timeout 600 python bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
timeout 1500 python -m pytest tests -m gpu -q --durations=15 -p no:cacheprovider --deselect tests/test_gpu_fullsize.py --deselect tests/test_gpu_nlb.py --deselect tests/test_gpu_dtensor.py > $OUT/pytest_rest.log 2>&1
echo "exit $?" >> $OUT/pytest_rest.log
tail -5 $OUT/pytest_new.log; tail -3 $OUT/pytest_rest.log; cat $OUT/bench.json | head -c 1500
