#!/bin/bash
# round 4, call 5: the general-rank LRA path (goldens at r = 96 / 130, fuzz up to r = 200), the 26-dim case judged against the fp64 oracle,
# and the library after the 256 x 128 tiling was taken out (seam tests)
OUT=gpurun_out/r04_call5
mkdir -p $OUT
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_lra.py tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider --timeout=200 -k "lra or 26" --durations=5 > $OUT/pytest_lra_26.log 2>&1; echo "exit $?" >> $OUT/pytest_lra_26.log
tail -40 $OUT/pytest_lra_26.log | cut -c1-400
timeout 300 python -m pytest tests/test_gpu_kron.py -m gpu -q -p no:cacheprovider --timeout=120 -x > $OUT/pytest_kron.log 2>&1; echo "exit $?" >> $OUT/pytest_kron.log
tail -4 $OUT/pytest_kron.log
for r in 96; do
  timeout 120 python bench.py --config vit-b-lra --lra-rank $r --steps 1 --warmup 1 --no-cpu-baseline --no-peaks > $OUT/bench_lra_r$r.json 2>> $OUT/bench.err
  python -c "import json;d=json.loads(open('$OUT/bench_lra_r$r.json').read().strip().splitlines()[-1]);print('lra rank $r', d['ms_per_step'], 'ms', d['roofline']['moved_gbs'], 'GB/s')"
done
timeout 200 python bench.py --config vit-b-lra --bf16 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_vit-b-lra_bf16.json 2>> $OUT/bench.err
python -c "import json;d=json.loads(open('$OUT/bench_vit-b-lra_bf16.json').read().strip().splitlines()[-1]);print('lra bf16', d['ms_per_step'], 'ms', d['roofline']['achieved'], d['roofline']['frac'])"
tail -5 $OUT/bench.err
