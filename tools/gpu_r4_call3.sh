#!/bin/bash
# round 4, call 3: row-split tensors on the real engine (two ranks, one GPU), the small-plan norm bound + fused diagonal update (LeNet5),
# the norm bound with half its slab requested before the start block (A/B), bench.py's new fields, then the whole GPU suite
OUT=gpurun_out/r04_call3
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_bench_multirank.py -m gpu -q -p no:cacheprovider > $OUT/pytest_sharded.log 2>&1; echo "exit $?" >> $OUT/pytest_sharded.log
tail -25 $OUT/pytest_sharded.log
python bench.py --config lenet5 --steps 200 --warmup 20 --no-cpu-baseline --no-apply-only --no-peaks > $OUT/bench_lenet5.json 2>> $OUT/bench.err
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/p_l5 -- python $R/bench.py --config lenet5 --steps 12 --warmup 4 --no-cpu-baseline --no-apply-only --no-peaks > /dev/null 2>> $R/$OUT/rocprof.err
  python $R/tools/rocpd_sequence.py $(find /tmp/p_l5 -name "*.db" | head -1) accumulate_kernel -3 > $R/$OUT/lenet5_step_sequence.md )
cat $OUT/lenet5_step_sequence.md | tail -22
for v in 0 1; do
  ( cd /tmp && PSGDK_NLB_EARLY=$v rocprofv3 --kernel-trace --stats -d /tmp/p_n$v -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-apply-only --no-peaks > $R/$OUT/bench_nlb_early$v.json 2>> $R/$OUT/rocprof.err
    python $R/tools/rocpd_sequence.py $(find /tmp/p_n$v -name "*.db" | head -1) accumulate_kernel -3 > $R/$OUT/step_sequence_nlb_early$v.md )
  echo "nlb early=$v"; grep "nlb_coop\|kernel time" $OUT/step_sequence_nlb_early$v.md
done
PSGDK_NLB_EARLY=1 timeout 300 python -m pytest tests/test_gpu_nlb.py tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -x -k "nlb or gpt2_small_shapes_bf16" > $OUT/pytest_nlb_early.log 2>&1; echo "exit $?" >> $OUT/pytest_nlb_early.log
tail -4 $OUT/pytest_nlb_early.log
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2>> $OUT/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_call3/bench.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("ms_per_step", "ms_per_step_median", "ms_per_step_min", "kernel_ms_per_step")})
print(d["config"].get("shader_clock_mhz_under_mfma_load"), d["roofline"].get("traffic_stale"), d["roofline"]["frac"], d["config"]["ranks_agree_bitwise"])
PY
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu_all.log 2>&1; echo "exit $?" >> $OUT/pytest_gpu_all.log
tail -12 $OUT/pytest_gpu_all.log
