#!/bin/bash
# round 6: cooperative norm bound for 1024-wide factors on plans that fit (a rank's share of GPT-2-medium): tests + rank arithmetic
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/r6z; mkdir -p $out
cd $R
timeout 1200 python -m pytest tests/test_gpu_nlb.py -x -q -m gpu > $out/pytest.log 2>&1; tail -3 $out/pytest.log
python tools/rank_arithmetic.py --world 8 --config gpt2-medium --chunks 1 --steps 30 --out $out/rank_gpt2-medium_c1.json > $out/rank_gpt2-medium_c1_k32.txt 2>&1; tail -2 $out/rank_gpt2-medium_c1_k32.txt
python tools/rank_arithmetic.py --world 8 --config gpt2-medium --chunks 2 --steps 30 --out $out/rank_gpt2-medium_c2.json > $out/rank_gpt2-medium_c2_k32.txt 2>&1; tail -2 $out/rank_gpt2-medium_c2_k32.txt
