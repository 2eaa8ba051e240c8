#!/bin/bash
# Collects the evidence profiles/ holds for one build.  Run ON THE GPU BOX from the repo root:
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r01_n'
# writes gpurun_out/<tag>/: bench lines of every config, the rocprofv3 kernel-trace summary + one step's dispatch sequence,
# HBM bytes per launch from two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined with other trace domains).
tag=${1:-latest}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
# (the headline line as the driver runs it: 20 steps after 5, secondary workloads in config.secondary; the traced runs below pass --no-secondary --
#  rocprofv3 follows the secondary workloads' child processes and would leave four result databases)
python $R/bench.py --steps 20 --warmup 5 2> $out/bench.err | tail -1 > $out/bench.json
python $R/bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline --no-peaks --no-secondary 2>> $out/bench.err | tail -1 > $out/bench_no_events.json
for c in gpt2-medium lenet5 gpt2-small-eq vit-b-lra; do
    python $R/bench.py --config $c --steps 30 --warmup 8 --no-cpu-baseline $( [ $c = vit-b-lra ] || echo --no-peaks ) 2>> $out/bench.err | tail -1 > $out/bench_$c.json
done
rocprofv3 --kernel-trace --stats -d /tmp/p_stats -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-apply-only --no-peaks --no-secondary > $out/bench_under_rocprof.json 2> $out/rocprof_stats.err
db=$(find /tmp/p_stats -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $db > $out/kernel_stats.md
python $R/tools/rocpd_sequence.py $db accumulate_kernel -3 > $out/step_sequence.md
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-apply-only --no-peaks --no-secondary > /dev/null 2> $out/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-apply-only --no-peaks --no-secondary > /dev/null 2> $out/pmc_write.err
python $R/tools/pmc_traffic.py $(find /tmp/p_fetch -name "*.db" | head -1) $(find /tmp/p_write -name "*.db" | head -1) > $out/pmc_traffic.json
ls -la $out
# SQ counters, two passes of four (MFMA pipe busy + LDS conflicts; where the wave cycles went) -- counters only with --kernel-trace
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/p_sq1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-apply-only --no-peaks --no-secondary > /dev/null 2> $out/pmc_sq1.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d /tmp/p_sq2 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-apply-only --no-peaks --no-secondary > /dev/null 2> $out/pmc_sq2.err
python $R/tools/pmc_sq.py $(find /tmp/p_sq1 -name "*.db" | head -1) > $out/pmc_sq_mfma.json
python $R/tools/pmc_sq.py $(find /tmp/p_sq2 -name "*.db" | head -1) > $out/pmc_sq_waits.json
# GPT-2-medium under rocprofv3 (the 8-GPU configuration's shapes on one GPU)
rocprofv3 --kernel-trace --stats -d /tmp/p_med -- python $R/bench.py --config gpt2-medium --steps 8 --warmup 3 --no-cpu-baseline --no-apply-only --no-peaks > $out/bench_gpt2-medium_under_rocprof.json 2> $out/rocprof_med.err
dbm=$(find /tmp/p_med -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $dbm > $out/gpt2-medium_kernel_stats.md
python $R/tools/rocpd_sequence.py $dbm accumulate_kernel -3 > $out/gpt2-medium_step_sequence.md
# the EQ geometry and the LRA path under rocprofv3
rocprofv3 --kernel-trace --stats -d /tmp/p_eq -- python $R/bench.py --config gpt2-small-eq --steps 8 --warmup 2 > /dev/null 2> $out/rocprof_eq.err
dbe=$(find /tmp/p_eq -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $dbe > $out/gpt2-small-eq_kernel_stats.md
rocprofv3 --kernel-trace --stats -d /tmp/p_lra -- python $R/bench.py --config vit-b-lra --steps 4 --warmup 1 > /dev/null 2> $out/rocprof_lra.err
dbl=$(find /tmp/p_lra -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $dbl > $out/vit-b-lra_kernel_stats.md
# (round 6) the bf16 leg of config 4 (SURVEY 8d quotes both dtypes) and an APPLY-ONLY step's dispatch sequence (update gated off: the steady state)
python $R/bench.py --config vit-b-lra --bf16 --steps 30 --warmup 8 --no-cpu-baseline 2>> $out/bench.err | tail -1 > $out/bench_vit-b-lra_bf16.json
rocprofv3 --kernel-trace --stats -d /tmp/p_lrab -- python $R/bench.py --config vit-b-lra --bf16 --steps 4 --warmup 1 > /dev/null 2> $out/rocprof_lra_bf16.err
python $R/tools/rocpd_stats.py $(find /tmp/p_lrab -name "*.db" | head -1) > $out/vit-b-lra_bf16_kernel_stats.md
rocprofv3 --kernel-trace --stats -d /tmp/p_ao -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-peaks --no-secondary --no-roofline > /dev/null 2> $out/rocprof_apply_only.err
python $R/tools/rocpd_sequence.py $(find /tmp/p_ao -name "*.db" | head -1) accumulate_kernel -3 > $out/apply_only_step_sequence.md
# LeNet5 dispatch sequence (config 2) and the HBM counters of the LRA passes (config 4)
rocprofv3 --kernel-trace --stats -d /tmp/p_l5 -- python $R/bench.py --config lenet5 --steps 12 --warmup 4 --no-cpu-baseline --no-apply-only --no-peaks > /dev/null 2> $out/rocprof_l5.err
python $R/tools/rocpd_sequence.py $(find /tmp/p_l5 -name "*.db" | head -1) accumulate_kernel -3 > $out/lenet5_step_sequence.md
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_lf -- python $R/bench.py --config vit-b-lra --steps 2 --warmup 1 --no-peaks > /dev/null 2> $out/pmc_lra.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_lw -- python $R/bench.py --config vit-b-lra --steps 2 --warmup 1 --no-peaks > /dev/null 2>> $out/pmc_lra.err
python $R/tools/pmc_traffic.py $(find /tmp/p_lf -name "*.db" | head -1) $(find /tmp/p_lw -name "*.db" | head -1) > $out/pmc_traffic_vit-b-lra.json
python $R/tests/parity_report.py > $out/parity_report.md 2> $out/parity_report.err
ls -la $out
