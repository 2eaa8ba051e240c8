#!/bin/bash
# round 6: cooperative norm bound with members of 128 columns (plans with few wide factors): soak test, stamps, a sharded rank's step
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r6t; mkdir -p $out
cd $R
timeout 1200 python -m pytest tests/test_gpu_nlb.py tests/test_gpu_fullsize.py -x -q -m gpu > $out/pytest.log 2>&1; tail -3 $out/pytest.log
for nv in 1 0; do
  PSGDK_NLB_NARROW=$nv python tools/nlb_stamps.py --width 768 --factors 8 --dtype bf16 2>&1 | grep -v amdgpu.ids > $out/stamps_8x768_narrow$nv.txt
  grep -A3 "chain 0" $out/stamps_8x768_narrow$nv.txt | head -3
  for cfg in gpt2-small gpt2-medium; do
    PSGDK_NLB_NARROW=$nv python tools/rank_arithmetic.py --world 8 --config $cfg --chunks 1 --steps 30 --out $out/rank_${cfg}_c1_narrow$nv.json > $out/rank_${cfg}_c1_narrow$nv.txt 2>&1
    tail -2 $out/rank_${cfg}_c1_narrow$nv.txt
  done
done
