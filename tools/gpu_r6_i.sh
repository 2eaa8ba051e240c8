#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
R=$(pwd); O=$R/gpurun_out/r6i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for dt in "" "--bf16"; do
  tag=fp32; [ -n "$dt" ] && tag=bf16
  rocprofv3 --kernel-trace --stats -d /tmp/p_l_$tag -- python $R/bench.py --config vit-b-lra $dt --steps 4 --warmup 2 --no-peaks > /dev/null 2> $O/stats_$tag.err
  python $R/tools/rocpd_stats.py $(find /tmp/p_l_$tag -name "*.db" | head -1) | head -14 > $O/stats_$tag.md
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d /tmp/p_w_$tag -- python $R/bench.py --config vit-b-lra $dt --steps 2 --warmup 1 --no-peaks > /dev/null 2> $O/pmc1_$tag.err
  python $R/tools/pmc_sq.py $(find /tmp/p_w_$tag -name "*.db" | head -1) > $O/pmc_waits_$tag.json
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d /tmp/p_v_$tag -- python $R/bench.py --config vit-b-lra $dt --steps 2 --warmup 1 --no-peaks > /dev/null 2> $O/pmc2_$tag.err
  python $R/tools/pmc_sq.py $(find /tmp/p_v_$tag -name "*.db" | head -1) > $O/pmc_valu_$tag.json
done
cat $O/stats_fp32.md $O/stats_bf16.md
python - <<'PY'
import json
for t in ('fp32','bf16'):
  for f in ('waits','valu'):
    try:
        d=json.load(open(f'/root/repo/gpurun_out/r6i/pmc_{f}_{t}.json'))['kernels']
        for k,v in d.items():
            if k.startswith('lra_') and 'small' not in k:
                c=v['counters']; wc=c['SQ_WAVE_CYCLES']
                print(t,f,k, 'us/disp', round(v['total_us']/v['dispatches']), {n: round(x/wc,3) for n,x in c.items() if n!='SQ_WAVE_CYCLES'})
    except Exception as e: print(t,f,'ERR',e)
PY
tail -3 $O/pmc2_bf16.err
