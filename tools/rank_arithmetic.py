#!/usr/bin/env python3
"""What ONE rank of an N-rank sharded job computes, timed on ONE MI355X (no node needed): the arithmetic side of SURVEY 8e's scaling question.

    python tools/rank_arithmetic.py --world 8 [--config gpt2-small] [--steps 30] [--out gpurun_out/rank_arithmetic.json]

For every rank k of the world the tool starts a fresh process that joins a FAKE process group of that size (torch's own
`torch.testing._internal.distributed.fake_pg`: every collective and every send / receive completes at once and moves nothing), builds the
sharded `KWNS4` exactly as `bench.py --gpus N` does -- same tensors, same owner map, same chunks, same row split -- and times its steps with the
exchange calls issued on the RCCL code path (device tensors, in-place gathers; `torch.distributed.get_backend` is patched to answer "nccl").
So a rank runs precisely the kernels it would run in the real job, in the same order, with NO time spent on the wire: max over ranks of the
result is the step time of an N-GPU job whose exchanges were free, i.e. the ceiling the arithmetic alone puts on strong scaling, measured instead
of modelled (DESIGN section 6 had only the cost model).  What the peers would have sent is absent (their segments of the exchange buffer stay zero):
parameters of tensors owned elsewhere do not move, the dense factor of a row-split tensor is fitted on this rank's rows only -- the numbers
are TIMES, the state is not a training state.  The exchanged bytes per step are printed beside the times for the wire model of section 6.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(args):
    import torch
    import torch.distributed as dist
    import bench
    world, rank = args.world, args.rank
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if world > 1:
        from torch.testing._internal.distributed.fake_pg import FakeStore
        dist.init_process_group(backend="fake", store=FakeStore(), rank=rank, world_size=world)
        dist.get_backend = lambda *a, **k: "nccl"          # the exchange code of the RCCL transport (device tensors, in-place gathers)
    import psgd_torch_amd
    shapes = bench.gpt2_shapes(n_layer=24, n_embd=1024) if args.config == "gpt2-medium" else bench.gpt2_shapes()
    gen = torch.Generator(device=dev).manual_seed(1234)
    params = [torch.nn.Parameter(0.02 * torch.randn(*s, device=dev, generator=gen)) for s in shapes]

    def synth_grad(shp):                                   # bench.py's structured gradients
        v = torch.randn(*shp, device=dev, generator=gen)
        if len(shp) != 2:
            return 0.01 * v
        m, n = shp
        sm = torch.logspace(0, -1.5, m, device=dev)[torch.randperm(m, device=dev, generator=gen)]
        sn = torch.logspace(0, -1.5, n, device=dev)[torch.randperm(n, device=dev, generator=gen)]
        g = sm[:, None] * v * sn[None, :]
        return g * (0.01 / g.square().mean().sqrt())
    grads = [[synth_grad(s) for s in shapes] for _ in range(2)]
    kw = dict(shard_state=True) if world > 1 else {}
    if args.no_row_split:
        kw["shard_split_rows"] = False
    if args.chunks:
        kw["shard_chunks"] = args.chunks
    opt = psgd_torch_amd.KWNS4(params, preconditioner_dtype=torch.bfloat16, **kw)

    def step(i):
        for p, g in zip(params, grads[i % 2]):
            p.grad = g
        opt.step()
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    prof = None
    if args.cprofile:
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    for i, (a, b) in enumerate(ev):
        a.record()
        step(i)
        b.record()
    enq = (time.perf_counter() - t0) / args.steps * 1e3          # the host's share: everything enqueued, nothing waited for
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / args.steps * 1e3
    if prof is not None:
        import io
        import pstats
        prof.disable()
        buf = io.StringIO()
        pstats.Stats(prof, stream=buf).sort_stats("tottime").print_stats(30)
        print(f"host profile of rank {rank} / {world} over {args.steps} steps (cProfile inflates python frames, not the time inside C calls):\n"
              + buf.getvalue(), file=sys.stderr, flush=True)
    ms = [a.elapsed_time(b) for a, b in ev]
    owned = split = 0
    sent = delivered = gathered = 0
    for b in opt._buckets.values():
        es = b.flat.element_size() if getattr(b, "flat", None) is not None else 0
        for (i, r0, r1, owner) in getattr(b, "pieces", []):
            owned += int(owner == rank)
        split += len(getattr(b, "blocks", {}))
        if es:
            sent += b.used[rank] * es                       # this rank's preconditioned gradients: what it has to send to every peer
            delivered += sum(b.used) * es                   # what every rank must end up with
            gathered += b.seg * world * es                  # what an equal-size all-gather of this chunk delivers (padding included)
    finite = all(bool(torch.isfinite(p).all()) for p in params)
    print("RESULT " + json.dumps(dict(rank=rank, world=world, ms_median=statistics.median(ms), ms_min=min(ms), ms_mean=sum(ms) / len(ms),
                                      wall_ms=wall, host_enqueue_ms=enq, pieces_owned=owned, row_split_tensors=split, bytes_sent=sent, bytes_needed=delivered, bytes_padded_gather=gathered,
                                      params_finite=finite)), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=-1, help="(internal) run this rank in this process")
    ap.add_argument("--config", default="gpt2-small", choices=["gpt2-small", "gpt2-medium"])
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--chunks", type=int, default=0)
    ap.add_argument("--no-row-split", action="store_true")
    ap.add_argument("--cprofile", action="store_true", help="cProfile the timed steps of the LAST rank (printed to stderr)")
    ap.add_argument("--ranks", default="", help="comma-separated subset of ranks to run (default: all)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    if args.rank >= 0:
        return child(args)
    rows = []
    for world in (1, args.world):
        for rank in (range(world) if (world == 1 or not args.ranks) else [int(x) for x in args.ranks.split(",")]):
            cmd = [sys.executable, os.path.abspath(__file__), "--world", str(world), "--rank", str(rank), "--config", args.config,
                   "--steps", str(args.steps), "--warmup", str(args.warmup)] + (["--chunks", str(args.chunks)] if args.chunks else []) \
                  + (["--no-row-split"] if args.no_row_split else []) + (["--cprofile"] if (args.cprofile and world > 1) else [])
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            if args.cprofile and world > 1:
                sys.stderr.write(r.stderr[r.stderr.find("host profile of rank"):] if "host profile of rank" in r.stderr else "")
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
            if r.returncode != 0 or not line:
                print(f"rank {rank} of {world} failed (exit {r.returncode}):\n{r.stderr[-2000:]}", file=sys.stderr)
                return 1
            rows.append(json.loads(line[0][7:]))
    single = rows[0]
    ranks = rows[1:]
    worst = max(r["ms_median"] for r in ranks)
    mean = sum(r["ms_median"] for r in ranks) / len(ranks)
    summary = dict(config=args.config, world=args.world, row_split=not args.no_row_split, single_ms_median=single["ms_median"],
                   slowest_rank_ms_median=worst, mean_rank_ms_median=mean, arithmetic_speedup_ceiling=single["ms_median"] / worst,
                   efficiency_ceiling=single["ms_median"] / worst / args.world, ranks=ranks, single=single)
    print("| rank | pieces owned | ms/step median | min | wall | host enqueue | MB sent to every peer | MB a rank needs |")
    print("|---:|---:|---:|---:|---:|---:|---:|---:|")
    print(f"| (1 GPU) | all | {single['ms_median']:.3f} | {single['ms_min']:.3f} | {single['wall_ms']:.3f} | {single['host_enqueue_ms']:.3f} | | |")
    for r in ranks:
        print(f"| {r['rank']} / {r['world']} | {r['pieces_owned']} | {r['ms_median']:.3f} | {r['ms_min']:.3f} | {r['wall_ms']:.3f} | {r['host_enqueue_ms']:.3f} | "
              f"{r['bytes_sent'] / 1e6:.1f} | {r['bytes_needed'] / 1e6:.1f} |")
    print(f"\nslowest rank {worst:.3f} ms, mean {mean:.3f} ms; 1 GPU {single['ms_median']:.3f} ms -> the arithmetic alone allows "
          f"{summary['arithmetic_speedup_ceiling']:.2f} x on {args.world} GPUs ({100 * summary['efficiency_ceiling']:.0f} % efficiency) "
          f"with row split {'on' if not args.no_row_split else 'off'}")
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(summary, f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
