#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r6k; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/rank_arithmetic.py --world 8 --config gpt2-small --steps 20 --chunks 1 --out $O/rank_small_c1.json > $O/rank_small_c1.txt 2>&1
timeout 900 python tools/rank_arithmetic.py --world 8 --config gpt2-small --steps 20 --out $O/rank_small.json > $O/rank_small.txt 2>&1
tail -15 $O/rank_small_c1.txt; tail -6 $O/rank_small.txt
