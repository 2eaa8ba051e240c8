#!/bin/bash
# round 4, calls 13 / 14 (HISTORICAL: the compile-time switch it used is gone, the winning mask is hard-coded in kernels_ew.hiph): non-temporal hints on the
# streaming passes (accumulate, emit); the variants were built on the CPU box as psgd_torch_amd/libpsgdk_nt<mask>.so with -DPSGDK_EW_NT=<mask> and run
# A/B/.../A on one box through tools/bench_with_lib.py.  Results: profiles/r04_i
OUT=$(pwd)/gpurun_out/r04_nt2
R=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in base nt15 nt7 nt31 nt3 nt11 base2; do
  lib=$R/psgd_torch_amd/libpsgdk_$v.so; [ $v = base ] && lib=$R/psgd_torch_amd/libpsgdk.so; [ $v = base2 ] && lib=$R/psgd_torch_amd/libpsgdk.so
  python $R/tools/bench_with_lib.py $lib --steps 30 --warmup 5 --no-cpu-baseline --no-peaks --no-apply-only 2> $OUT/bench_$v.err | tail -1 > $OUT/bench_$v.json
  rm -rf /tmp/p_$v
  rocprofv3 --kernel-trace --stats -d /tmp/p_$v -- python $R/tools/bench_with_lib.py $lib --steps 12 --warmup 4 --no-cpu-baseline --no-apply-only --no-peaks --no-roofline > /dev/null 2> $OUT/rocprof_$v.err
  python $R/tools/rocpd_stats.py $(find /tmp/p_$v -name "*.db" | head -1) > $OUT/kernel_stats_$v.md
  python - <<PY
import json,re
d=json.loads(open("$OUT/bench_$v.json").read().strip().splitlines()[-1])
ks=open("$OUT/kernel_stats_$v.md").read()
def avg(name):
    m=re.search(r"\`"+name+r"[^|]*\| *(\d+) *\| *([\d.]+) *\| *([\d.]+)", ks)
    return float(m.group(3)) if m else None
print("$v", "median", round(d["ms_per_step_median"],4), "min", round(d["ms_per_step_min"],4), "mean", round(d["ms_per_step"],4),
      "| accumulate", avg("_Z17accumulate_kernelIt"), "emit", avg("_Z11emit_kernelIt"), "pipe", avg("_Z19gemm_nt_pipe_kernelIt"), "gemm128", avg("_Z14gemm_nt_kernelItLb0"))
PY
done
