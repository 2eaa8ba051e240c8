#!/bin/bash
# round 6 A/B inside one call: ab_libs/lib_base.so (HEAD) against the working tree's libpsgdk.so; base first and last
#   gpurun -- 'bash tools/gpu_r6_ab.sh <tag> [pytest files...]'
tag=${1:-ab}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
if [ $# -gt 0 ]; then
  timeout 1500 python -m pytest "$@" -x -q -m gpu > $out/pytest.log 2>&1; tail -4 $out/pytest.log
fi
show() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d.get("roofline") or {}
print(sys.argv[1].split("/")[-1], "mean", round(d["ms_per_step"], 4), "median", round(d.get("ms_per_step_median") or 0, 4), "min", round(d.get("ms_per_step_min") or 0, 4),
      "gemm", round(r.get("gemm_ms_per_step") or 0, 4), "apply_only", d.get("config", {}).get("apply_only_ms_per_step"))
PY
}
for v in base new base new; do
  lib=$R/psgd_torch_amd/libpsgdk.so; [ $v = base ] && lib=$R/ab_libs/lib_base.so
  for c in gpt2-small lenet5; do
    n=$(ls $out | grep -c "bench_${c}_${v}")
    python tools/bench_with_lib.py $lib --config $c --steps 30 --warmup 5 --no-cpu-baseline --no-peaks --no-secondary 2>> $out/bench.err | tail -1 > $out/bench_${c}_${v}_$n.json
    show $out/bench_${c}_${v}_$n.json
  done
done
