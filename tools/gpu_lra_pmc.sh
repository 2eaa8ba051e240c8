#!/bin/bash
# SQ counters of the LRA passes (two passes of four counters; never combined with other trace domains)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/p_l -- python $R/bench.py --config vit-b-lra --steps 2 --warmup 1 > /dev/null 2>&1
python $R/tools/pmc_sq.py $(find /tmp/p_l -name "*.db" | head -1) > $R/gpurun_out/pmc_sq_lra.json
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d /tmp/p_l2 -- python $R/bench.py --config vit-b-lra --steps 2 --warmup 1 > /dev/null 2>&1
python $R/tools/pmc_sq.py $(find /tmp/p_l2 -name "*.db" | head -1) > $R/gpurun_out/pmc_sq_lra2.json
