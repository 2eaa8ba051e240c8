#!/usr/bin/env python3
"""bench.py -- PSGD-Kron step() throughput on GPT-2-small parameter shapes (BASELINE.json's metric and config).

A "step" is ONE full KWNS4.step() over all 148 parameter tensors of GPT-2-small (124,475,904 params): coupled work
of wrapped_as_torch_optimizer_for_ddp.py:98-176 -- momentum EMA, preconditioner update (probability 1) with the
Q0.5EQ1.5 geometry, preconditioning, clipping, parameter update -- with bf16 preconditioner state, fp32 parameters and
synthetic fp32 gradients already resident in HBM.  N > 1: either preconditioner state sharded per parameter across the
ranks (shard_state=True; the clipped preconditioned gradients are exchanged with one all-gather) or plain replicas --
--parallelism auto (default) times the sharded mode (chunked all-gathers, one all-gather, chunked point-to-point) and plain replicas in warm-up and keeps the fastest; total work is fixed, so scaling is "strong".

Prints ONE JSON line (rank 0).  Besides the driver's contract it carries
  roofline     -- for the dominant kernel (the grouped NT MFMA GEMM, both tilings): the algorithmic FLOPs of a step that run in
                  it (SURVEY 8d model: 905.2 GFLOP per step, minus the 2 x 256 d^2 of the two norm bounds when the cooperative
                  kernel runs them = 886.5 GFLOP in 9 launches) / their launch time measured live with hipEvents on the launch
                  stream, against the 2.5 PFLOP/s dense bf16 MFMA peak;
                  `peak_measured` holds this chip's own ceilings (MFMA loops of both bf16 shapes on register operands, a streaming
                  copy and read), measured in-process after the timed region;
  cpu_baseline -- the CPU oracle (a port, test infrastructure) timed on this box's host cores on the WHOLE workload (all 148
                  tensors), at 8 threads and at the physical cores of one socket.
The synthetic gradients are the structured variant of SURVEY 8d (g = H1 V H2) unless --gaussian-grads.
After the timed region every parameter and the whole preconditioner state are checked to be finite (a fast wrong step is not
a measurement).
"""
import argparse
import json
import math
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on these hosts (RCCL / multi-process)

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def lib_sha256():
    """Hash of the built engine library: stamps the PMC traffic profile (tools/pmc_traffic.py writes it) so that bench.py can tell
    whether roofline.traffic was measured on the code that is running."""
    import hashlib
    path = os.path.join(ROOT, "psgd_torch_amd", "libpsgdk.so")
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def gpt2_shapes(n_layer=12, n_embd=768, vocab=50304, block=1024):
    """misc/gpt2.py: GPTConfig defaults (gpt2.py:215-227), tied wte/lm_head (gpt2.py:252-254)."""
    s = [(vocab, n_embd), (block, n_embd)]
    for _ in range(n_layer):
        s += [(n_embd,), (n_embd,), (3 * n_embd, n_embd), (3 * n_embd,), (n_embd, n_embd), (n_embd,),
              (n_embd,), (n_embd,), (4 * n_embd, n_embd), (4 * n_embd,), (n_embd, 4 * n_embd), (n_embd,)]
    s += [(n_embd,), (n_embd,)]
    return s


def flop_model(shapes, max_skew=1.0, nlb_in_gemm=True):
    """SURVEY 8d.  Returns (step FLOPs, FLOPs that run in the grouped GEMM kernel).  The subspace iterations of the two norm
    bounds (2 x 4 products of a 32 x d block: 512 d^2) are grouped-GEMM problems only on the multi-launch route; when the
    cooperative kernel runs them (Engine.info()["nlb_coop"]) they are not part of the GEMM launches that are timed."""
    step = gemm = 0.0
    for shp in shapes:
        N = math.prod(shp)
        for d in shp:
            if d <= 1 or d * d > max_skew * N:
                continue
            ap = min(4.0 * N * d, 2.0 * d ** 3 + 2.0 * N * d)
            step += 2 * ap + 2.0 * N * d + 6.0 * d ** 3 + 512.0 * d ** 2
            gemm += 2 * ap + 2.0 * N * d + 6.0 * d ** 3 + (512.0 * d ** 2 if nlb_in_gemm else 0.0)
    return step, gemm


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def _physical_cores_of_one_socket():
    """(physical cores of socket 0, number of sockets) from /proc/cpuinfo; (None, None) if it cannot be read."""
    try:
        phys, cores = set(), {}
        cur = {}
        for line in list(open("/proc/cpuinfo")) + ["\n"]:
            if line.strip() == "":
                if "physical id" in cur:
                    phys.add(cur["physical id"])
                    cores.setdefault(cur["physical id"], set()).add(cur.get("core id", len(cores.get(cur["physical id"], ()))))
                cur = {}
                continue
            k, _, v = line.partition(":")
            cur[k.strip()] = v.strip()
        if not phys:
            return None, None
        return len(cores[sorted(phys)[0]]), len(phys)
    except OSError:
        return None, None


def cpu_baseline(block_budget=2.5):
    """The CPU oracle (oracle/psgd_oracle.py, a port of the reference path, test infrastructure) timed on this box's host cores on
    the WHOLE workload: the identical KWNS4.step over all 148 GPT-2-small tensors (124,475,904 parameters, wte included), bf16
    preconditioner, same hyper-parameters, synthetic gradients -- SURVEY 8d.  Two thread counts: 8 (comparable with the survey
    container's probe of the reference itself) and the physical cores of one socket; one untimed step (state
    initialisation) and then 3 / 2 timed steps, about 10-25 s of CPU work in all.  `value` is the faster of the two (`cores` says
    which).  A short sample of ONE transformer block is kept as a secondary field (comparable with rounds 1-2)."""
    from oracle import psgd_oracle as orc
    shapes_all = gpt2_shapes()
    n_all = torch.get_num_threads()
    socket_cores, n_sockets = _physical_cores_of_one_socket()
    if not socket_cores:
        socket_cores = n_all

    def run(shapes, threads, timed_steps, budget=None):
        torch.set_num_threads(threads)
        gen = torch.Generator().manual_seed(0)
        params = [0.02 * torch.randn(*s, generator=gen) for s in shapes]
        opt = orc.KWNS4Oracle(params, seed=0)
        times = []
        t_all = time.time()
        for it in range(1 + timed_steps):
            grads = [0.01 * torch.randn(*s, generator=gen) for s in shapes]
            t0 = time.time()
            opt.step(grads)
            times.append(time.time() - t0)
            if budget is not None and it >= 2 and time.time() - t_all > budget:
                break
        t = sorted(times[1:])
        return t[len(t) // 2], len(t)
    nparam = sum(math.prod(s) for s in shapes_all)
    full_step_flops, _ = flop_model(shapes_all)
    try:
        s8, k8 = run(shapes_all, min(8, n_all), 3)
        ss, ks = (s8, k8) if socket_cores == min(8, n_all) else run(shapes_all, min(socket_cores, n_all), 2)
        blk = shapes_all[2:14]
        b8, kb8 = run(blk, min(8, n_all), 30, budget=block_budget)
    finally:
        torch.set_num_threads(n_all)
    best, cores = (s8, min(8, n_all)) if s8 <= ss else (ss, min(socket_cores, n_all))
    blk_flops, _ = flop_model(blk)
    return {"value": nparam / best / 1e9, "unit": "Gparam/s", "cores": cores, "kind": "port", "cpu_model": _cpu_model(),
            "s_per_step": best, "gflops": full_step_flops / best / 1e9,
            "threads8": {"value": nparam / s8 / 1e9, "s_per_step": s8, "gflops": full_step_flops / s8 / 1e9, "cores": min(8, n_all), "timed_steps": k8},
            "one_socket": {"value": nparam / ss / 1e9, "s_per_step": ss, "gflops": full_step_flops / ss / 1e9,
                           "cores": min(socket_cores, n_all), "sockets_on_box": n_sockets, "timed_steps": ks},
            "block_sample_threads8": {"ms_per_block_step": b8 * 1e3, "gflops": blk_flops / b8 / 1e9, "timed_steps": kb8,
                                      "what": "ONE of the 12 transformer blocks (12 tensors), as rounds 1-2 timed it"},
            "sample": f"the WHOLE GPT-2-small step: KWNS4Oracle.step over all 148 tensors ({nparam} params, {full_step_flops / 1e9:.0f} GFLOP), "
                      f"bf16 preconditioner; median of {k8} steps at 8 threads ({s8:.2f} s) and of {ks} steps at {min(socket_cores, n_all)} threads = the "
                      f"physical cores of one socket ({ss:.2f} s), one untimed initialisation step before each; baseline only"}


def bench_lra(args):
    """BASELINE config 4: ViT-B/16 parameter count (N = 86,543,080), LRA rank 10, fp32: update_precond_lra_whiten +
    precond_grad_lra per step (psgd.py:1066, 1055).  HBM-bound: algorithmic bytes = 9 matrix reads + 3 matrix writes
    (update) + 3 matrix reads (apply), matrix pass = N*r*4 bytes, plus 18 + 3 N-vector passes (the kernels
    actually move 30 vector passes, DESIGN.md section 3 has the per-pass table)."""
    from psgd_torch_amd import lra
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    N, r = 86543080, args.lra_rank
    dt_ = torch.bfloat16 if args.bf16 else torch.float32
    esz = 2 if args.bf16 else 4
    gen = torch.Generator(device=dev).manual_seed(0)
    U = torch.randn(N, r, device=dev, generator=gen); U *= 0.1 ** 0.5 / torch.linalg.vector_norm(U)
    V = torch.randn(N, r, device=dev, generator=gen); V *= 0.1 ** 0.5 / torch.linalg.vector_norm(V)
    UVd = [U.to(dt_), V.to(dt_), torch.ones(N, 1, device=dev, dtype=dt_)]
    del U, V
    Luvd = [torch.zeros([], device=dev) for _ in range(3)]
    g = (0.01 * torch.randn(N, 1, device=dev, generator=gen)).to(dt_)

    def one_step():
        lra.update_precond_lra_whiten(UVd, Luvd, g, lr=0.1, betaL=0.9, damping=1e-9)
        return lra.precond_grad_lra(UVd, g)
    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / args.steps
    packed = UVd[2]._psgdk_lra.info()["packed_rows"]          # psgdk_lra_info: which row kernels the last call took
    row_kernels = (f"packed two-rows-per-thread passes (csrc/kernels_lra_pk.hiph) over {packed} rows, one-row kernels over the last {N - packed}"
                   if packed else "one-row kernels (csrc/kernels_lra.hiph)")
    bytes_alg = (12 + 3) * N * r * esz + (18 + 3) * N * esz    # 9 R + 3 W + 3 R matrix passes, 18 + 3 N-vector passes
    # what the kernels move: 24 vector passes since round 3; since round 6 the Grams of psgd.py:1006 are carried from
    # update to update and the factors re-read for them every lra.GRAM_EVERY updates only: 2 of the 12 + 3 matrix passes run once in 16 updates
    gram_passes = 2.0 / lra.GRAM_EVERY if (lra.GRAM_EVERY and r <= 64) else 2.0
    bytes_moved = (10 + gram_passes + 3) * N * r * esz + 24 * N * esz
    peaks = None
    if not args.no_peaks:
        import ctypes as C
        from psgd_torch_amd import _lib
        scratch = torch.empty(2 << 30, dtype=torch.uint8, device=dev)
        scratch.random_(0, 255)
        pk = (C.c_float * 4)()
        _lib.check(_lib.probe_lib().psgdk_test_peaks(pk, scratch.data_ptr(), scratch.numel(), _lib.current_stream()), "test_peaks")
        del scratch
        peaks = {"hbm_copy_gbs": pk[2], "hbm_read_gbs": pk[3],
                 "what": "streaming 16-byte copy (read + write) / read of 1 GiB, best of 3, measured in this process after the timed region"}
    traffic = None          # HBM bytes per update + apply from the separate rocprofv3 --pmc passes (profiles/), all LRA launches summed
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic_vit-b-lra_latest.json")
    if os.path.exists(tpath) and r == 10 and not args.bf16:
        try:
            tj = json.load(open(tpath))["kernels"]
            per_step = {}
            for k, v in tj.items():
                if k.startswith("lra_"):
                    per_step[k] = v["hbm_bytes_per_launch_corrected"] * v["dispatches_traced"]
            steps_traced = max(1, tj["lra_gram_kernelIf"]["dispatches_traced"]) if "lra_gram_kernelIf" in tj else 3
            traffic = sum(per_step.values()) / steps_traced
        except Exception:
            traffic = None
    out = {"metric": "psgd_lra_update_apply_throughput", "value": N / dt / 1e9, "unit": "Gparam/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "bf16" if args.bf16 else "fp32", "data": "synthetic",
           "config": {"workload": f"ViT-B/16 parameter count N={N}, LRA rank {r}, {'bf16' if args.bf16 else 'fp32'}: update_precond_lra_whiten + precond_grad_lra",
                      "rank": r,
                      "row_kernels": row_kernels,
                      "true_gram_passes_in_timed_region": (len([k for k in range(args.warmup, args.warmup + args.steps) if k % lra.GRAM_EVERY == 0])
                                                           if (lra.GRAM_EVERY and r <= 64) else args.steps),
                      "gram_recurrence": (f"the Grams of psgd.py:1006 carried from update to update (r x r recurrence), the factors re-read for them every "
                                          f"{lra.GRAM_EVERY} updates: 13 + 2/{lra.GRAM_EVERY} matrix passes per update + apply instead of 15; `achieved` prices the "
                                          "SURVEY's 15 (algorithmic bytes of the path as the reference defines it), `moved_gbs` what the kernels moved")
                      if (lra.GRAM_EVERY and r <= 64) else None},
           "roofline": {"bound": "hbm", "achieved": bytes_alg / dt / 1e9, "peak": 8000.0, "unit": "GB/s",
                        "frac": bytes_alg / dt / 1e9 / 8000.0, "traffic": traffic,
                        "traffic_note": ("counters collected with a Gram pass in every update (rounds 2-5); since round 6 two of the fifteen matrix "
                                         "passes run once in 16 updates: moved_gb_per_step is the current figure") if traffic else None,
                        "traffic_source": ("profiles/pmc_traffic_vit-b-lra_latest.json: separate rocprofv3 --pmc passes (2 x FETCH_SIZE + WRITE_SIZE), "
                                           "all psgdk_lra_* launches of one update + apply") if traffic else None,
                        "algorithmic_gb_per_step": bytes_alg / 1e9,
                        "moved_gb_per_step": bytes_moved / 1e9, "moved_gbs": bytes_moved / dt / 1e9, "peak_measured": peaks,
                        "moved_frac_of_measured_read": (bytes_moved / dt / 1e9 / peaks["hbm_read_gbs"]) if peaks else None}}
    print(json.dumps(out), flush=True)


def bench_eq(args):
    """The triangular geometry (psgd.py:278-336) on the headline shapes: per step accumulate ->
    psgdk_update_precond_eq -> precond_grad -> clipped parameter update, driven through KronEngine like KronWhiten does."""
    from psgd_torch_amd import KronEngine, _lib as L
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    shapes = gpt2_shapes()
    nparam = sum(math.prod(s) for s in shapes)
    gen = torch.Generator(device=dev).manual_seed(1234)
    params = [0.02 * torch.randn(*s, device=dev, generator=gen) for s in shapes]
    grads = [[0.01 * torch.randn(*s, device=dev, generator=gen) for s in shapes] for _ in range(2)]
    eng = KronEngine(shapes, dev, precond_dtype=torch.bfloat16, max_skew=1.0, use_momentum=True, init_scale=1.0, geometry="EQ")

    def one_step(i):
        # (the momentum pass also writes the update's probe V and damped input, as KWNS4 / KronWhiten ask it to)
        eng.accumulate(grads[i % 2], beta=0.9, keep_grad=False, damp=dict(source=L.SRC_EMA, damping=1e-9, seed=1, offset=i))
        eng.update_precond(L.SRC_EMA, 0.1, 0.9, 1e-9, seed=1, offset=i, balance_mask=None)
        eng.precond_grad(L.SRC_EMA)
        eng.apply_update(params, 2e-4, 0.0, 2.0, 10.0)
    for i in range(args.warmup):
        one_step(i)
    eng.profile_read(reset=True)
    eng.profile_enable(False)
    sample = 10 if args.steps >= 20 else (4 if args.steps >= 8 else 1)            # (event pairs around the GEMM launches on every 4th step only: see main())
    prof_steps = 0
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        on = (not args.no_roofline) and (i % sample == sample - 1)
        prof_steps += int(on)
        eng.profile_enable(on)
        one_step(args.warmup + i)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / args.steps
    gemm_ms, gemm_launches = eng.profile_read(reset=True)
    eng.profile_enable(False)
    # FLOPs of the products that run in the grouped GEMM kernel, per dense factor d of a tensor of N elements (psgd.py:278-327):
    # A = (kron Q) Hvp 2Nd, the two mode Grams 2 x 2Nd, Q -= mu triu(.) Q 2d^3, apply: P = Q^T Q 2d^3 + h = (kron P) g 2Nd.
    # `dense`: as the reference's einsums count them (they do not exploit the triangular Q); `executed`: with the zero halves of
    # triangular operands skipped (Q in A: half of K; triu(.) Q: a third of the d^3 cube) -- the roofline uses the SMALLER one.
    dense = executed = 0.0
    for shp in shapes:
        N = math.prod(shp)
        for d in shp:
            if d <= 1 or d * d > N:
                continue
            dense += 8.0 * N * d + 4.0 * d ** 3
            executed += 7.0 * N * d + (2.0 / 3.0 + 2.0) * d ** 3
    out = {"metric": "psgd_kron_step_throughput", "value": nparam / dt / 1e9, "unit": "Gparam/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "bf16", "data": "synthetic",
           "config": {"workload": f"GPT-2-small parameter shapes ({nparam} params), dQ=EQ (triangular Q, fp32 right solves), "
                                  "momentum 0.9, update probability 1, max_skew 1", "preconditioner_dtype": "bf16"}}
    if prof_steps and gemm_launches:
        per_step_ms = gemm_ms / prof_steps
        ach = executed / (per_step_ms * 1e-3) / 1e12
        out["roofline"] = {"bound": "mfma", "kernel": "gemm_nt_kernel + gemm_nt_pipe_kernel <bf16> (all grouped-GEMM launches of the EQ step; "
                                                      "the triangular solves run in eq_trsm_bf16_kernel and are not part of it)",
                           "achieved": ach, "peak": 2500.0, "unit": "TFLOP/s", "frac": ach / 2500.0, "traffic": None,
                           "launches_per_step": gemm_launches / prof_steps, "avg_launch_us": gemm_ms * 1e3 / gemm_launches,
                           "algorithmic_gflop_per_launch": executed / 1e9 / (gemm_launches / prof_steps),
                           "gemm_ms_per_step": per_step_ms, "gflop_dense_as_reference": dense / 1e9, "gflop_executed": executed / 1e9}
    print(json.dumps(out), flush=True)


def secondary_workloads():
    """BASELINE.json's other single-GPU workloads (configs 2, 4, 5 as a 1-GPU workload), each a few steps in a process of its own after the
    headline's timed region (this process still holds the headline's state; the GPU is otherwise idle): median device ms per step and
    the workload's own roofline fraction, so that the driver's record carries all of them.  Not part of `value`."""
    import subprocess
    res = {}
    # (LRA: 16 timed steps after 5 = one true Gram pass among them -- lra.GRAM_EVERY = 16 -- so the figure carries the recurrence's amortised cost)
    for name, extra, steps in (("lenet5", [], 200), ("vit-b-lra", [], 16), ("vit-b-lra-bf16", ["--bf16"], 16), ("gpt2-medium", [], 10)):
        cmd = [sys.executable, os.path.abspath(__file__), "--config", name.replace("-bf16", ""), "--steps", str(steps), "--warmup", "5", "--no-cpu-baseline",
               "--no-secondary", "--no-apply-only", "--no-peaks"] + extra
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
            line = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1]
            d = json.loads(line)
            rf = d.get("roofline") or {}
            res[name] = {"workload": d["config"].get("workload"), "ms_per_step": d["ms_per_step"], "ms_per_step_median": d.get("ms_per_step_median"),
                         "value": d["value"], "unit": d["unit"], "steps": d["steps"], "dtype": d["dtype"],
                         "roofline_bound": rf.get("bound"), "roofline_frac": rf.get("frac"), "roofline_achieved": rf.get("achieved"),
                         "roofline_unit": rf.get("unit"), "whole_step_frac_of_peak": rf.get("whole_step_frac_of_peak"),
                         "launches_per_step": d["config"].get("launches_per_step"),
                         "host_enqueue_ms_per_step": d["config"].get("host_enqueue_ms_per_step"), "wall_s": time.perf_counter() - t0}
        except Exception as e:      # noqa: BLE001  (a secondary figure must never take the headline line down)
            res[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--whiten-grad", action="store_true", help="variant: fit P on the gradient, apply it to the momentum "
                                                               "(KWNS4(whiten_grad=True)); the headline uses the default False")
    ap.add_argument("--no-roofline", action="store_true", help="do not time the GEMM launches with hipEvents (no roofline "
                                                               "object; shows what the event pairs cost the step)")
    ap.add_argument("--no-apply-only", action="store_true", help="skip the secondary apply-only measurement (profiling runs)")
    ap.add_argument("--gaussian-grads", action="store_true", help="white-noise gradients 0.01 N(0,1) instead of the structured "
                                                                  "g = H1 V H2 (SPD H, condition ~1e3: SURVEY 8d) the headline uses, "
                                                                  "under which the preconditioner actually moves while timed")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads (BASELINE configs 2, 4, 5) that the default "
                                                                "single-GPU run times after the headline and reports under config.secondary")
    ap.add_argument("--no-peaks", action="store_true", help="skip the in-process ceiling measurement (roofline.peak_measured)")
    ap.add_argument("--no-fuse", action="store_true", help="A/B: the parameter update as its own streaming pass (psgdk_apply_update) instead of "
                    "fused into the epilogue of the apply's last product (psgdk_precond_grad_apply)")
    ap.add_argument("--fuse-stagger", type=int, default=-1, help="experiment: GemmUpdArgs::stagger of the fused launches in 100 MHz ticks "
                    "(-1: the library's default)")
    ap.add_argument("--fp32", action="store_true", help="fp32 preconditioner instead of bf16 (not the headline config)")
    ap.add_argument("--bf16", action="store_true", help="vit-b-lra only: bf16 factors and vectors instead of fp32 (SURVEY 8d quotes both)")
    ap.add_argument("--config", default="gpt2-small", choices=["gpt2-small", "gpt2-medium", "lenet5", "vit-b-lra", "gpt2-small-eq"],
                    help="BASELINE.json configs; the default (gpt2-small) is the headline metric's configuration")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to smoke-test "
                                                      "the multi-rank code path with --same-device)")
    ap.add_argument("--lra-rank", type=int, default=10, help="vit-b-lra only: rank of the approximation (BASELINE config 4: 10)")
    ap.add_argument("--same-device", action="store_true", help="testing only: every rank uses cuda:0")
    ap.add_argument("--parallelism", default="auto", choices=["auto", "sharded", "replicated"],
                    help="N > 1: per-parameter state sharding + one all-gather per step, or replicas (the reference's DDP "
                         "semantics, no exchange); auto times both during warm-up and keeps the faster one")
    args = ap.parse_args()
    if args.config == "vit-b-lra":
        return bench_lra(args)
    if args.config == "gpt2-small-eq":
        return bench_eq(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP engine)")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = world > 1
    if dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            torch.distributed.init_process_group(backend="nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend=args.backend)

    import psgd_torch_amd
    if args.config == "gpt2-medium":
        shapes = gpt2_shapes(n_layer=24, n_embd=1024)
    elif args.config == "lenet5":
        shapes = [(6, 26), (16, 151), (257, 120), (121, 84), (85, 10)]      # mnist_with_lenet5.py:20-24
        args.fp32 = True
    else:
        shapes = gpt2_shapes()
    nparam = sum(math.prod(s) for s in shapes)
    gen = torch.Generator(device=dev).manual_seed(1234)
    params = [torch.nn.Parameter(0.02 * torch.randn(*s, device=dev, generator=gen)) for s in shapes]
    pd = torch.float32 if args.fp32 else torch.bfloat16
    # synthetic gradient streams resident in HBM: a few distinct draws, cycled
    n_sets = 2

    def synth_grad(shp):
        v = torch.randn(*shp, device=dev, generator=gen)
        if args.gaussian_grads or len(shp) != 2:
            return 0.01 * v
        # g = H1 V H2 with diagonal-in-a-random-basis-free SPD factors: per-row and per-column scales spread log-uniformly over
        # 1.5 decades each (condition ~1e3 of H1 (x) H2), shuffled; normalised to the same RMS as the white-noise variant
        m, n = shp
        sm = torch.logspace(0, -1.5, m, device=dev)[torch.randperm(m, device=dev, generator=gen)]
        sn = torch.logspace(0, -1.5, n, device=dev)[torch.randperm(n, device=dev, generator=gen)]
        g = sm[:, None] * v * sn[None, :]
        return g * (0.01 / g.square().mean().sqrt())
    grad_sets = [[synth_grad(s) for s in shapes] for _ in range(n_sets)]

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    MODES = {"sharded": dict(shard_state=True, shard_chunks=2),         # 2 chunks: the first chunk's all-gather under the second one's arithmetic
             # (named explicitly since round 6: an optimizer that names no count now takes 1 or 2 from the exchange model -- the probe below times both)
             "sharded, one exchange": dict(shard_state=True, shard_chunks=1),
             "sharded, four chunks": dict(shard_state=True, shard_chunks=4),
             "sharded, p2p": dict(shard_state=True, shard_exchange="p2p"),       # every chunk as 2 (N - 1) direct sends / receives

             "replicated": dict(shard_state=False), "single": dict(shard_state=False)}

    def make(mode_name):
        ps = [torch.nn.Parameter(p.detach().clone()) for p in params]
        o = psgd_torch_amd.KWNS4(ps, preconditioner_dtype=pd, whiten_grad=args.whiten_grad,
                                 **MODES[mode_name])                                       # reference defaults otherwise
        o._fuse_update = not args.no_fuse
        return ps, o

    def step_of(ps, o):
        def f(i):
            gs = grad_sets[i % n_sets]
            for p, g in zip(ps, gs):
                p.grad = g
            o.step()
        return f

    # N > 1: the path shards per parameter (state + compute on the owner, ONE all-gather of the clipped preconditioned
    # gradients per step) -- worthwhile when the exchange costs less than the compute it saves, which depends on N and the
    # fabric.  Both modes give every rank the same parameters; auto keeps the faster one (timed here, untimed warm-up).
    mode = "sharded" if dist else "single"
    if dist and args.parallelism != "sharded":
        mode = "replicated"
    if dist and args.parallelism == "auto":
        timing, failures = {}, []
        for name in ("sharded", "sharded, one exchange", "sharded, four chunks", "sharded, p2p", "replicated"):
            # (a mode that raises -- e.g. a collective the installed RCCL / torch refuses -- is dropped from the probe on every rank
            #  alike: argument errors are deterministic; the timed region then runs with what is left)
            try:
                ps_, o_ = make(name)
                f_ = step_of(ps_, o_)
                for i in range(3):
                    f_(i)
                sync_all()
                t_ = time.perf_counter()
                for i in range(4):
                    f_(i)
                sync_all()
                el = time.perf_counter() - t_
                del ps_, o_, f_
            except Exception as e:      # noqa: BLE001
                print(f"bench: rank {rank}: mode {name!r} failed in the probe: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
                failures.append(f"{name}: {type(e).__name__}: {e}")
                el = float("inf")
            # the failed flag travels with the time, in the FIRST collective after the try block: a failure on one rank only (out of
            # memory, a transport error) marks the mode as failed everywhere
            tt = torch.tensor([el if math.isfinite(el) else 1e300], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            timing[name] = float(tt.item()) / 4 if float(tt.item()) < 1e299 else float("inf")
            torch.cuda.empty_cache()
        if not any(math.isfinite(v) for v in timing.values()):
            raise SystemExit("bench.py: every parallelism mode failed in the warm-up probe: " + "; ".join(failures or ["(on another rank)"]))
        mode = min(timing, key=timing.get)
    params, opt = make(mode)
    one_step = step_of(params, opt)

    # everything that takes host time without keeping the GPU busy happens BEFORE the warm-up steps (collector run, event objects, the
    # engines' profiling switches): the timed region then follows the W warm-up steps after nothing but the barrier + synchronize the
    # contract asks for.  (Rounds 1-4 ran the collector and built 21 event objects between warm-up and timed region: tens of
    # milliseconds of idle GPU, after which the first six to ten timed steps ran 5 - 40 % slow while the clocks came back --
    # profiles/r05_a: step times 2.56, 1.88, 1.92, 1.86, 1.85, 1.80, 1.76, 1.73 ... ms.)
    import gc
    gc_ms = [0.0]
    gc_t = [0.0]

    def gc_watch(phase, info):
        if phase == "start":
            gc_t[0] = time.perf_counter()
        else:
            gc_ms[0] += (time.perf_counter() - gc_t[0]) * 1e3
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    one_step(0)                                   # (builds the buckets / engines; counted as the first warm-up step)
    engines = [b.engine for b in opt._buckets.values() if b.engine is not None]
    for e in engines:
        e.profile_read(reset=True)
        e.profile_enable(False)
        if args.fuse_stagger >= 0:
            e.fuse_update(not args.no_fuse, args.fuse_stagger)
    gc.collect()
    gc_before = [g["collections"] for g in gc.get_stats()]
    gc.callbacks.append(gc_watch)
    # step 0 built the engines; at least one more untimed step follows the collector run, so that the timed region never starts on an
    # idle GPU and never repeats a step index (with --warmup 0 / 1 that is one / no step more than asked: reported as warmup_steps_run)
    n_warm = max(args.warmup, 2)
    for i in range(1, n_warm):
        one_step(i)
    # the grouped-GEMM launches of every `sample`-th timed step carry an event pair each (for the roofline object).  The events ride on
    # the launches' own dispatch packets (hipExtLaunchKernelGGL) and no longer fence the stream; the four hot-path calls' own pairs do,
    # so not every step is sampled.  The roofline's launch durations are those steps' launches, measured live on the launch stream
    # inside the timed region.
    sample = 4 if args.steps >= 8 else 1
    prof_steps = 0

    fence = sync_all
    # one event per step boundary on the stream the engine launches on (torch's current stream): no fences, nothing waits on them
    # until the timed region is over; the per-step device times give the median / min next to the wall-clock mean
    sampled = []
    import ctypes as _C
    from psgd_torch_amd import _lib as _L
    _nl = _C.c_int64()
    fence()
    _L.check(_L.lib().psgdk_test_launch_count(_C.byref(_nl), 1), "launch_count")      # (reset: the timed region's launches are counted)
    t0 = time.perf_counter()
    step_ev[0].record()
    for i in range(args.steps):
        on = (not args.no_roofline) and world == 1 and (i % sample == sample - 1)
        if on:
            prof_steps += 1
        sampled.append(on)
        for e in engines:
            e.profile_enable(on, calls=on and prof_steps == 1)      # (the calls' own event pairs fence the stream: one sampled step carries them)
        one_step(n_warm + i)
        step_ev[i + 1].record()
    host_dt = time.perf_counter() - t0          # host enqueue time (no sync inside): shows whether the host keeps ahead
    _L.check(_L.lib().psgdk_test_launch_count(_C.byref(_nl), 0), "launch_count")
    kernel_launches_per_step = _nl.value / args.steps
    fence()
    dt = time.perf_counter() - t0
    per_step = [step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(args.steps)]
    plain = sorted(x for x, on in zip(per_step, sampled) if not on) or sorted(per_step)      # steps without the roofline event pairs
    step_median, step_min = plain[len(plain) // 2], plain[0]
    gc.callbacks.remove(gc_watch)
    gc_in_timed = {"collections_by_generation": [g["collections"] - b for g, b in zip(gc.get_stats(), gc_before)],
                   "ms_per_step": gc_ms[0] / args.steps}
    gemm_ms, gemm_launches, call_ms, fused_ms, fused_launches = 0.0, 0, 0.0, 0.0, 0
    for e in engines:
        fm, fn = e.profile_read_fused()
        fused_ms += fm
        fused_launches += fn
        ms, n = e.profile_read(reset=True)
        gemm_ms += ms
        gemm_launches += n
        call_ms += e.profile_read_calls(reset=True)[0]
        e.profile_enable(False)
    if dist:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())

    # a fast wrong step is not a measurement: every parameter and the whole preconditioner state (Q, Q^T, diagonal factors,
    # L, momentum: one arena per engine) must be finite after the timed region
    bad = [i for i, p in enumerate(params) if not bool(torch.isfinite(p).all())]
    for e in engines:
        for t in range(e.n):
            for q in e.Q[t]:
                if not bool(torch.isfinite(q.float()).all()):
                    bad.append(("Q", t))
            for ell in e.Lip[t]:
                if not bool(torch.isfinite(ell).all()):
                    bad.append(("L", t))
            if e.ema[t] is not None and not bool(torch.isfinite(e.ema[t].float()).all()):
                bad.append(("ema", t))
    if bad:
        raise SystemExit(f"bench.py: non-finite values after the timed region: {bad[:8]}")
    nlb_fallbacks = sum(e.info()["nlb_fallbacks"] for e in engines)
    # every rank must hold the same parameters after the timed region: a checksum that any differing bit changes, compared over the ranks
    chk = torch.zeros(2, dtype=torch.float64, device=dev)
    for k, p_ in enumerate(params):
        bits = p_.detach().view(torch.int32).to(torch.float64)
        chk[0] += bits.sum() * (1 + (k % 7))
        chk[1] += (bits * bits).sum() * 1e-12
    ranks_agree = True
    if dist:
        lo, hi = chk.clone(), chk.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        ranks_agree = bool(torch.equal(lo, hi))
        # sharded modes: every rank applies the SAME gathered h to its parameters -- any difference is a bug.  Replicas (the reference's
        # DDP semantics) recompute everything with atomics whose order is free: they agree to rounding, not bit for bit (hence the
        # reference's resync_every)
        if not ranks_agree and mode.startswith("sharded"):
            raise SystemExit(f"bench.py: rank {rank}: the ranks' parameters differ after the timed region (checksum {chk.tolist()})")

    # secondary figure: the apply-only step, i.e. the steady state once the update probability is annealed down
    # (momentum + precondition + clip + parameter update; the preconditioner update gated off).  Outside the timed region.
    apply_only_ms = None
    if not dist and not args.no_apply_only:
        for g in opt.param_groups:
            g["preconditioner_update_probability"] = 1e-12
        for i in range(3):
            one_step(i)
        fence()
        t1 = time.perf_counter()
        for i in range(10):
            one_step(i)
        fence()
        apply_only_ms = (time.perf_counter() - t1) / 10 * 1e3

    ms_per_step = dt / args.steps * 1e3
    nlb_coop = bool(engines) and all(e.info()["nlb_coop"] for e in engines)
    step_flops, gemm_flops = flop_model(shapes, nlb_in_gemm=not nlb_coop)
    run_cpu = args.config == "gpt2-small"
    out = {
        "metric": "psgd_kron_step_throughput",
        "value": nparam / (dt / args.steps) / 1e9,
        "unit": "Gparam/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        # device time per step from one event per step boundary (steps that do not carry the roofline's event pairs): the wall-clock mean
        # above moves with the host and the box by 2-4 %, the median / min of the device times do not
        "ms_per_step_median": step_median, "ms_per_step_min": step_min,
        # sum over the engine's hot-path calls of (first kernel start -> last kernel end), on the sampled steps: the step without the host
        "kernel_ms_per_step": call_ms if prof_steps else None,      # (the one sampled step whose hot-path calls carried event pairs)
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "fp32" if args.fp32 else "bf16",
        "data": "synthetic" + (" (white-noise gradients)" if args.gaussian_grads else " (structured gradients g = H1 V H2, SURVEY 8d)"),
        "config": {"workload": {"gpt2-small": "GPT-2-small parameter shapes (misc/gpt2.py GPTConfig defaults): 148 tensors, "
                                              f"{nparam} params, 62 dense 768x768 Kron factors",
                                "gpt2-medium": f"GPT-2-medium parameter shapes (24 layers, d=1024): {len(shapes)} tensors, {nparam} params",
                                "lenet5": f"LeNet5 parameter shapes (mnist_with_lenet5.py): 5 tensors, {nparam} params, fp32"}[args.config]
                               + ("; KWNS4 defaults (momentum 0.9, whiten momentum, update probability 1, max_skew 1)" if not args.whiten_grad
                                  else "; KWNS4 defaults except whiten_grad=True (momentum 0.9, update probability 1, max_skew 1)"),
                   "preconditioner_dtype": "fp32" if args.fp32 else "bf16", "param_dtype": "fp32",
                   "parallelism": "single GPU" if world == 1 else ({"sharded": f"per-parameter state sharding x{world}, all-gathers of 2 chunks overlapped with the arithmetic",
                                                                    "sharded, one exchange": f"per-parameter state sharding x{world} + one all-gather per step",
                                                                    "sharded, four chunks": f"per-parameter state sharding x{world}, all-gathers of 4 chunks overlapped with the arithmetic",
                                                                    "sharded, p2p": f"per-parameter state sharding x{world}, 2 chunks exchanged by direct point-to-point sends"}.get(
                                                                       mode, f"replicas x{world} (no exchange step)")),
                   "parallelism_probe_ms": ({k: (v * 1e3 if math.isfinite(v) else None) for k, v in timing.items()}
                                            if (dist and args.parallelism == "auto") else None),
                   "step_gflop_model": step_flops / 1e9, "host_enqueue_ms_per_step": host_dt / args.steps * 1e3,
                   "launches_per_step": kernel_launches_per_step,      # every kernel the library launched in the timed region / steps
                   "param_update": ("fused into the epilogue of the apply's last product (psgdk_precond_grad_apply)" if not (args.no_fuse or dist)
                                    else "separate streaming pass"),
                   "step_device_ms": [round(x, 4) for x in per_step],      # every timed step, in order (event to event on the launch stream)
                   "gc_in_timed_region": gc_in_timed, "apply_only_ms_per_step": apply_only_ms,
                   "norm_bound_route": "cooperative launch (device-scope exchange)" if nlb_coop else "grouped-GEMM products",
                   "norm_bound_timeouts": nlb_fallbacks, "state_finite_after_timed_region": True,
                   "ranks_agree_bitwise": ranks_agree, "param_checksum": [float(x) for x in chk.tolist()]},
    }
    if world == 1 and gemm_launches and prof_steps:
        launches_per_step = gemm_launches / prof_steps
        # FLOPs of the launches that carry the fused update = the apply's products S P (apply = 2 d^3 + 2 N d per dense factor, of
        # which 2 d^3 is P = Q^T Q in a launch of its own)
        fused_flops = float(sum(2.0 * math.prod(s) * d for s in shapes for d in s if d > 1 and d * d <= math.prod(s))) if fused_launches else 0.0
        avg_launch_s = gemm_ms / 1e3 / gemm_launches
        achieved = (gemm_flops / launches_per_step) / avg_launch_s / 1e12
        peak = 157.3 if args.fp32 else 2500.0
        # HBM bytes per launch come from a separate rocprofv3 --pmc run (profiles/): counters cannot be read in-process
        traffic = traffic_lib = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")
        if os.path.exists(tpath) and not args.fp32 and args.config == "gpt2-small":
            try:
                tj = json.load(open(tpath))
                traffic = (tj.get("gemm_all_tilings") or tj["kernels"]["gemm_nt_kernelIt"])["hbm_bytes_per_launch_corrected"]
                traffic_lib = tj.get("library_sha256")
            except Exception:
                traffic = None
        out["roofline"] = {"bound": "mfma" if args.config != "lenet5" else "latency",      # (LeNet5 is launch-latency bound: us / step and launches / step are its figures)
                           "kernel": "gemm_nt_kernel + gemm_nt_pipe_kernel <bf16> (all grouped-GEMM launches of the step: the "
                                                      "128 x 128 tiling and the persistent 256 x 256 one)"
                           if not args.fp32 else ("gemm_nt_kernel<float>" if args.config != "lenet5" else "gemm_nt_ks_kernel<float>"),
                           "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                           "traffic": traffic,
                           # the counters were collected on the library whose hash the profile carries; a different library now = stale
                           "traffic_library_sha256": traffic_lib if traffic is not None else None,
                           "traffic_stale": (traffic_lib != lib_sha256()) if traffic is not None else None,
                           "traffic_source": ("profiles/pmc_traffic_latest.json: separate rocprofv3 --pmc passes of this command on this "
                                              "code (FETCH_SIZE x 2 + WRITE_SIZE per launch); counters cannot be read in-process")
                           if traffic is not None else None,
                           "launches_per_step": launches_per_step,
                           "avg_launch_us": avg_launch_s * 1e6,
                           "algorithmic_gflop_per_launch": gemm_flops / launches_per_step / 1e9,
                           "gemm_ms_per_step": gemm_ms / prof_steps, "steps_with_events": prof_steps,
                           # round 6: the apply's last product carries the parameter update in its epilogue (10 B / parameter of HBM traffic that
                           # the separate streaming pass of rounds 1-5 moved): that launch is slower than the bare product, the step faster.
                           # `achieved` / `frac` above price ALL grouped-GEMM launches, the fused one included, with the products' FLOPs only.
                           "fused_update_launch": ({"launches_per_step": fused_launches / prof_steps, "ms_per_step": fused_ms / prof_steps,
                                                    "what": "gemm_nt_pipe_kernel carrying p <- p (1 - wd lr) - lr clamp(h) in its epilogue: the product's "
                                                            "FLOPs plus 10 B / parameter (h out, fp32 parameter in and out)",
                                                    "hbm_gbs_algorithmic": (nparam * 12.0 / 1e9) / (fused_ms / prof_steps * 1e-3) if fused_ms > 0 else None}
                                                   if fused_launches else None),
                           "frac_excluding_fused_update_launch": (((gemm_flops - fused_flops) / 1e12) / ((gemm_ms - fused_ms) / prof_steps * 1e-3) / peak
                                                                  if fused_launches and gemm_ms > fused_ms else None),
                           "whole_step_frac_of_peak": step_flops / (dt / args.steps) / 1e12 / peak}
    if world == 1 and "roofline" in out and not args.no_peaks:
        # ceilings of THIS chip, measured in-process in < 1 s (SURVEY 8d "re-verify on the box"): register-operand MFMA loops of both
        # bf16 shapes, a streaming copy and a streaming read of 1 GiB
        import ctypes as C
        from psgd_torch_amd import _lib
        scratch = torch.empty(2 << 30, dtype=torch.uint8, device=dev)
        scratch.random_(0, 255)
        pk = (C.c_float * 4)()
        _lib.check(_lib.probe_lib().psgdk_test_peaks(pk, scratch.data_ptr(), scratch.numel(), _lib.current_stream()), "test_peaks")
        del scratch
        clk = C.c_float()
        _lib.check(_lib.probe_lib().psgdk_test_clock(C.byref(clk), _lib.current_stream()), "test_clock")
        out["config"]["shader_clock_mhz_under_mfma_load"] = clk.value
        r = out["roofline"]
        r["peak_measured"] = {"mfma_16x16x32_bf16_tflops": pk[0], "mfma_32x32x16_bf16_tflops": pk[1], "hbm_copy_gbs": pk[2],
                              "hbm_read_gbs": pk[3],
                              "what": "register-operand MFMA loops (2 waves per SIMD, 16 / 4 independent accumulator chains) and a "
                                      "streaming 16-byte copy / read of 1 GiB, best of 3, measured in this process after the timed region"}
        if not args.fp32:
            r["frac_of_measured_mfma"] = r["achieved"] / max(pk[0], pk[1])
            r["whole_step_frac_of_measured_mfma"] = r["whole_step_frac_of_peak"] * peak / max(pk[0], pk[1])
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and run_cpu:
            out["cpu_baseline"] = cpu_baseline()
        if world == 1 and run_cpu and not args.no_secondary and not args.fp32 and not args.whiten_grad:
            out["config"]["secondary"] = secondary_workloads()
        print(json.dumps(out), flush=True)
    if dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
