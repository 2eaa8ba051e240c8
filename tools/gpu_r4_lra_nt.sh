#!/bin/bash
# round 4, call 16 (HISTORICAL, product untouched): non-temporal loads (1) / stores (2) on the LRA row passes' factor streams, built from a scratch copy of csrc/ as
# psgd_torch_amd/libpsgdk_lra<mask>.so; ViT-B r = 10, A/B/.../A on one box.  Results: profiles/r04_i/README.md
OUT=$(pwd)/gpurun_out/r04_lra_nt
R=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in base lra1 lra3 lra2 base2; do
  lib=$R/psgd_torch_amd/libpsgdk_$v.so; [ $v = base ] && lib=$R/psgd_torch_amd/libpsgdk.so; [ $v = base2 ] && lib=$R/psgd_torch_amd/libpsgdk.so
  python $R/tools/bench_with_lib.py $lib --config vit-b-lra --steps 20 --warmup 5 --no-cpu-baseline --no-peaks 2> $OUT/bench_$v.err | tail -1 > $OUT/bench_$v.json
  python -c "import json;d=json.loads(open('$OUT/bench_$v.json').read().strip().splitlines()[-1]);print('$v fp32', round(d['ms_per_step'],3))"
done
for v in base lra3; do
  lib=$R/psgd_torch_amd/libpsgdk_$v.so; [ $v = base ] && lib=$R/psgd_torch_amd/libpsgdk.so
  python $R/tools/bench_with_lib.py $lib --config vit-b-lra --bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-peaks 2> $OUT/bench_bf16_$v.err | tail -1 > $OUT/bench_bf16_$v.json
  python -c "import json;d=json.loads(open('$OUT/bench_bf16_$v.json').read().strip().splitlines()[-1]);print('$v bf16', round(d['ms_per_step'],3))"
done
