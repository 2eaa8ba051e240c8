#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the LRA and EQ configurations, SQ counters of the EQ configuration
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/${1:-extra}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for cfg in vit-b-lra gpt2-small-eq; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/x_f_$cfg -- python $R/bench.py --config $cfg --steps 2 --warmup 1 > /dev/null 2> $out/pmc_fetch_$cfg.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/x_w_$cfg -- python $R/bench.py --config $cfg --steps 2 --warmup 1 > /dev/null 2> $out/pmc_write_$cfg.err
  python $R/tools/pmc_traffic.py $(find /tmp/x_f_$cfg -name "*.db" | head -1) $(find /tmp/x_w_$cfg -name "*.db" | head -1) > $out/pmc_traffic_$cfg.json
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/x_sq1 -- python $R/bench.py --config gpt2-small-eq --steps 2 --warmup 1 > /dev/null 2> $out/pmc_sq1_eq.err
python $R/tools/pmc_sq.py $(find /tmp/x_sq1 -name "*.db" | head -1) > $out/pmc_sq_mfma_gpt2-small-eq.json
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d /tmp/x_sq2 -- python $R/bench.py --config gpt2-small-eq --steps 2 --warmup 1 > /dev/null 2> $out/pmc_sq2_eq.err
python $R/tools/pmc_sq.py $(find /tmp/x_sq2 -name "*.db" | head -1) > $out/pmc_sq_waits_gpt2-small-eq.json
ls -la $out
