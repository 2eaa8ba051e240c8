#!/bin/bash
# The 1/2/4/8-GPU table of the headline config on ONE node, one command (the first run on a real node):
#   tools/scale.sh [config] [steps]        e.g.  tools/scale.sh gpt2-medium 30
# For every N it runs bench.py the way the driver does (torch.distributed.run, one rank per GPU over RCCL) and prints
# N, ms/step, Gparam/s, speed-up over N = 1, the mode --parallelism auto chose and the three probe times.
cd "$(dirname "$0")/.." || exit 1
CFG=${1:-gpt2-small}; STEPS=${2:-30}
export HSA_ENABLE_IPC_MODE_LEGACY=0
NG=$(python -c 'import torch; print(torch.cuda.device_count())')
mkdir -p gpurun_out/scale
for N in 1 2 4 8; do
  [ "$N" -gt "$NG" ] && break
  if [ "$N" -eq 1 ]; then
    python bench.py --gpus 1 --steps "$STEPS" --warmup 5 --config "$CFG" --no-cpu-baseline > gpurun_out/scale/n$N.json 2> gpurun_out/scale/n$N.err
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((29500 + N)) \
      bench.py --gpus "$N" --steps "$STEPS" --warmup 5 --config "$CFG" > gpurun_out/scale/n$N.json 2> gpurun_out/scale/n$N.err
  fi
done
python - <<'PY'
import glob, json
rows = {}
for f in sorted(glob.glob("gpurun_out/scale/n*.json")):
    for ln in open(f):
        if ln.startswith("{"):
            d = json.loads(ln); rows[d["n_gpus"]] = d
base = rows.get(1, {}).get("ms_per_step")
print(f"{'N':>2s} {'ms/step':>9s} {'Gparam/s':>9s} {'speed-up':>8s}  mode / probe (ms per step)")
for n, d in sorted(rows.items()):
    su = f"{base / d['ms_per_step']:.2f}x" if base else "-"
    print(f"{n:2d} {d['ms_per_step']:9.3f} {d['value']:9.1f} {su:>8s}  {d['config']['parallelism']} / {d['config'].get('parallelism_probe_ms')}")
PY
