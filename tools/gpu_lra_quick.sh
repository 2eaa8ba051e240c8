#!/bin/bash
# quick LRA round: golden + fuzz parity, bench at ranks 10 / 32 / 64 (no full-size oracle runs)
out=gpurun_out/${1:-lraq}; mkdir -p $out
PSGDK_FUZZ_LRA=200 timeout 900 python -m pytest tests/test_gpu_lra.py tests/test_gpu_fuzz.py -x -q -k "lra" 2>&1 | tail -3
for r in 10 32 64; do
  python bench.py --config vit-b-lra --lra-rank $r --steps 5 --warmup 2 2>/dev/null > $out/bench_lra_r$r.json
  python -c "import json; z=json.load(open('$out/bench_lra_r$r.json')); print($r, z['ms_per_step'], z['roofline']['achieved'])"
done
