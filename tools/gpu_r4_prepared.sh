#!/bin/bash
# First GPU call of the next round: the five experiments that were written after round 3's GPU budget was spent (nothing in the default path
# uses them; tools/isa_fingerprint.py check shows the production kernels unchanged).  ~9 minutes of box time.
#   1. tools/nlb_stamps.py -- where the 67 us of a cooperative norm bound go (instrumented instantiation, psgdk_test_nlb_stamps)
#   2. PSGDK_GEMM_KSPLIT=1 -- 64 x 64 tiles with the K loop split over the waves for stages of few tiles (gemm_nt_ks_kernel): parity on every
#      small-plan test, then LeNet5's step with and without it, same box
OUT=gpurun_out/r04_prepared
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
# (each experiment under its own timeout: blind code)
timeout 60 python tools/nlb_stamps.py --width 768 --factors 62 --dtype bf16 > $OUT/nlb_stamps_bf16_768.txt 2>&1; echo "exit $?" >> $OUT/nlb_stamps_bf16_768.txt
timeout 60 python tools/nlb_stamps.py --width 320 --factors 24 --dtype fp32 > $OUT/nlb_stamps_fp32_320.txt 2>&1; echo "exit $?" >> $OUT/nlb_stamps_fp32_320.txt
timeout 60 python tools/nlb_stamps.py --width 128 --factors 5 --dtype fp32 > $OUT/nlb_stamps_fp32_128_solo.txt 2>&1; echo "exit $?" >> $OUT/nlb_stamps_fp32_128_solo.txt
# K-split tiles: the GEMM kernel test itself does not go through plan_bind; everything below does
PSGDK_GEMM_KSPLIT=1 timeout 300 python -m pytest tests/test_gpu_kron.py tests/test_gpu_eq.py -m gpu -q -p no:cacheprovider -x \
    -k "functional_seam or kwns4_step or kronwhiten or known_answer or degenerate or eq" > $OUT/pytest_ksplit.log 2>&1; echo "exit $?" >> $OUT/pytest_ksplit.log
PSGDK_GEMM_KSPLIT=1 timeout 200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider -x > $OUT/pytest_ksplit_fuzz.log 2>&1; echo "exit $?" >> $OUT/pytest_ksplit_fuzz.log
for v in 0 1; do
  PSGDK_GEMM_KSPLIT=$v python bench.py --config lenet5 --steps 200 --warmup 20 --no-cpu-baseline --no-apply-only --no-peaks > $OUT/bench_lenet5_ksplit$v.json 2>> $OUT/bench.err
  ( cd /tmp && PSGDK_GEMM_KSPLIT=$v rocprofv3 --kernel-trace --stats -d /tmp/p_k$v -- python $R/bench.py --config lenet5 --steps 12 --warmup 4 --no-cpu-baseline --no-apply-only --no-peaks > /dev/null 2>> $R/$OUT/rocprof.err
    db=$(find /tmp/p_k$v -name "*.db" | head -1); python $R/tools/rocpd_sequence.py $db accumulate_kernel -3 > $R/$OUT/lenet5_step_sequence_ksplit$v.md )
done
# 3. PSGDK_ACC_EARLY_EMA=1 -- the momentum pass with the EMA loads issued before the noise chain (accumulate_kernel<T, true>): parity, then
#    the GPT-2-small step with and without it on this box (dispatch sequence: accumulate_kernel's duration)
#    (=2: plus the straight-line path for interior tiles -- gradient, parameter and EMA groups requested back to back, noise while they fly)
for v in 1 2; do
PSGDK_ACC_EARLY_EMA=$v timeout 300 python -m pytest tests/test_gpu_kron.py tests/test_gpu_production_path.py -m gpu -q -p no:cacheprovider -x \
    -k "kwns4_step or kronwhiten or functional_seam or fp32 or small_full_plan and True" > $OUT/pytest_early_ema$v.log 2>&1; echo "exit $?" >> $OUT/pytest_early_ema$v.log
done
for v in 0 1 2; do
  ( cd /tmp && PSGDK_ACC_EARLY_EMA=$v rocprofv3 --kernel-trace --stats -d /tmp/p_e$v -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-apply-only --no-peaks > $R/$OUT/bench_early_ema$v.json 2>> $R/$OUT/rocprof.err
    db=$(find /tmp/p_e$v -name "*.db" | head -1); python $R/tools/rocpd_sequence.py $db accumulate_kernel -3 > $R/$OUT/step_sequence_early_ema$v.md )
done
# 4. PSGDK_PIPE_SLACK_US -- the persistent 256 x 256 launch with the workgroups that walk one tile fewer started late (gemm_nt_pipe_kernel<T, true>):
#    bit-identical outputs (same tiles, same order per workgroup), so one parity run; then the two full-size products' durations at several delays
PSGDK_PIPE_SLACK_US=14 timeout 200 python -m pytest tests/test_gpu_production_path.py -m gpu -q -p no:cacheprovider -x -k "small_full_plan and True" > $OUT/pytest_slack.log 2>&1; echo "exit $?" >> $OUT/pytest_slack.log
for us in 0 7 14 21; do
  ( cd /tmp && PSGDK_PIPE_SLACK_US=$us rocprofv3 --kernel-trace --stats -d /tmp/p_s$us -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-apply-only --no-peaks > $R/$OUT/bench_slack$us.json 2>> $R/$OUT/rocprof.err
    db=$(find /tmp/p_s$us -name "*.db" | head -1); python $R/tools/rocpd_sequence.py $db accumulate_kernel -3 > $R/$OUT/step_sequence_slack$us.md )
  echo slack_us=$us; grep gemm_nt_pipe $OUT/step_sequence_slack$us.md; tail -1 $OUT/step_sequence_slack$us.md
done
# 5. PSGDK_LRA_EARLY_VEC=1 -- the LRA row passes with the block's N-vector elements requested before the next block's matrix prefetch
#    (EV instantiations): parity on the LRA suite, then ViT-B r = 10 with and without it, per-kernel durations
PSGDK_LRA_EARLY_VEC=1 timeout 300 python -m pytest tests/test_gpu_lra.py tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -x -k "lra and not true_n" > $OUT/pytest_lra_ev.log 2>&1; echo "exit $?" >> $OUT/pytest_lra_ev.log
for v in 0 1; do
  ( cd /tmp && PSGDK_LRA_EARLY_VEC=$v rocprofv3 --kernel-trace --stats -d /tmp/p_l$v -- python $R/bench.py --config vit-b-lra --steps 5 --warmup 2 --no-cpu-baseline --no-peaks > $R/$OUT/bench_lra_ev$v.json 2>> $R/$OUT/rocprof.err
    db=$(find /tmp/p_l$v -name "*.db" | head -1); python $R/tools/rocpd_stats.py $db > $R/$OUT/lra_kernel_stats_ev$v.md )
  echo lra_early_vec=$v; grep "lra_" $OUT/lra_kernel_stats_ev$v.md | head -8
done
tail -3 $OUT/pytest_lra_ev.log
tail -3 $OUT/pytest_slack.log
tail -3 $OUT/pytest_early_ema1.log; tail -3 $OUT/pytest_early_ema2.log; for v in 0 1 2; do echo early_ema=$v; grep accumulate $OUT/step_sequence_early_ema$v.md; tail -1 $OUT/step_sequence_early_ema$v.md; done
tail -25 $OUT/nlb_stamps_bf16_768.txt; tail -3 $OUT/pytest_ksplit.log; tail -3 $OUT/pytest_ksplit_fuzz.log
python - <<'EOF'
import json
for v in (0, 1):
    try:
        d = json.loads(open(f"gpurun_out/r04_prepared/bench_lenet5_ksplit{v}.json").read().strip().splitlines()[-1])
        print("lenet5 ksplit", v, round(d["ms_per_step"], 4), "ms/step")
    except Exception as e:
        print("lenet5 ksplit", v, "failed:", e)
EOF
