#!/bin/bash
# LRA round on the GPU box: parity of all rank classes, then the ViT-B bench at r = 10 (regression check) and r = 32 / 64
tag=${1:-lra}
out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_lra.py tests/test_gpu_fuzz.py -x -q -k "lra" > $out/pytest_lra.log 2>&1; echo "exit $?" >> $out/pytest_lra.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "lra" > $out/pytest_lra_full.log 2>&1; echo "exit $?" >> $out/pytest_lra_full.log
python bench.py --config vit-b-lra --steps 10 --warmup 3 > $out/bench_lra_r10.json 2> $out/bench_lra_r10.err
for r in 32 64; do
  python bench.py --config vit-b-lra --lra-rank $r --steps 5 --warmup 2 > $out/bench_lra_r$r.json 2> $out/bench_lra_r$r.err
done
tail -3 $out/pytest_lra.log $out/pytest_lra_full.log
for f in $out/bench_lra_r*.json; do python - "$f" <<'P'
import json, sys
try:
    z = json.load(open(sys.argv[1])); print(sys.argv[1], z["ms_per_step"], z["roofline"]["achieved"], z["config"].get("rank"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
P
done
