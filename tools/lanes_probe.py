"""Round-6 experiment: a GPT-2-small KWNS4 step as K independent LANES on K HIP streams.

The tensors of an optimizer are independent units, so the step can be cut into K cost-balanced sets, each with its own
engine (plan + arenas), launched on its own stream: lane A's latency-bound launches (norm bounds, rsub, the small diagonal kernels:
~0.23 ms of a 1.6 ms step that leave most CUs idle) can then run beside lane B's grouped GEMMs and streaming passes.  This script
measures it with NO library change: K KWNS4 instances over a partition of the parameter list, stepped back to back under
torch.cuda.stream(lane_k).  Run on the GPU box:   python tools/lanes_probe.py [--steps 30]
"""
import argparse
import re
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402
import psgd_torch_amd  # noqa: E402
from psgd_torch_amd.sharding import kron_step_cost  # noqa: E402


def partition(shapes, k, mode):
    costs = [kron_step_cost(tuple(s), float("inf"), 1.0) for s in shapes]
    if mode.startswith("big"):
        # lane 1 = the n largest tensors (+ with "d": every diagonal-only tensor); lane 0 = the rest
        n = int(re.sub(r"[^0-9]", "", mode) or 1)
        order = sorted(range(len(shapes)), key=lambda i: -costs[i])
        b = set(order[:n])
        if mode.endswith("d"):
            b |= {i for i, s in enumerate(shapes) if len(s) < 2}
        lanes = [[i for i in range(len(shapes)) if i not in b], sorted(b)]
        return lanes, [sum(costs[i] for i in ln) for ln in lanes]
    if mode.startswith("frac"):
        # lane 1 = a fraction of the cost, taken from every tensor class alike (every m-th tensor in cost order)
        f = float(mode[4:])
        order = sorted(range(len(shapes)), key=lambda i: -costs[i])
        lanes, load, acc = [[], []], [0.0, 0.0], 0.0
        tot = sum(costs)
        for i in order:
            j = 1 if load[1] + costs[i] <= f * tot and acc >= 1.0 else 0
            acc = acc - 1.0 + f / (1 - f) if j else acc + f / (1 - f)
            lanes[j].append(i)
            load[j] += costs[i]
        return [sorted(x) for x in lanes], load
    order = sorted(range(len(shapes)), key=lambda i: -costs[i])
    load = [0.0] * k
    lanes = [[] for _ in range(k)]
    for i in order:
        j = min(range(k), key=lambda r: load[r])
        lanes[j].append(i)
        load[j] += costs[i]
    for ln in lanes:
        ln.sort()
    return lanes, load


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--lanes", type=int, nargs="*", default=[1, 2, 3, 4])
    ap.add_argument("--config", default="gpt2-small")
    ap.add_argument("--modes", nargs="*", default=["lpt"])
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    shapes = bench.gpt2_shapes() if args.config == "gpt2-small" else bench.gpt2_shapes(n_layer=24, n_embd=1024)
    gen = torch.Generator(device=dev).manual_seed(1234)
    base = [0.02 * torch.randn(*s, device=dev, generator=gen) for s in shapes]

    def synth(shp):
        v = torch.randn(*shp, device=dev, generator=gen)
        if len(shp) != 2:
            return 0.01 * v
        m, n = shp
        sm = torch.logspace(0, -1.5, m, device=dev)[torch.randperm(m, device=dev, generator=gen)]
        sn = torch.logspace(0, -1.5, n, device=dev)[torch.randperm(n, device=dev, generator=gen)]
        g = sm[:, None] * v * sn[None, :]
        return g * (0.01 / g.square().mean().sqrt())
    grads = [[synth(s) for s in shapes] for _ in range(2)]

    for k, mode in [(k, m) for k in args.lanes for m in (args.modes if k == 2 else ["lpt"])]:
        lanes, load = partition(shapes, k, mode.rstrip("R"))
        if mode.endswith("R"):      # host enqueue order: the small lane first
            lanes, load = lanes[::-1], load[::-1]
        params = [torch.nn.Parameter(p.clone()) for p in base]
        opts = [psgd_torch_amd.KWNS4([params[i] for i in ln], preconditioner_dtype=torch.bfloat16) for ln in lanes]
        streams = [torch.cuda.Stream(device=dev) for _ in range(k)] if k > 1 else [torch.cuda.current_stream(dev)]
        main_stream = torch.cuda.current_stream(dev)

        def step(i):
            gs = grads[i % 2]
            for p, g in zip(params, gs):
                p.grad = g
            if k == 1:
                opts[0].step()
                return
            fork = torch.cuda.Event()
            fork.record(main_stream)
            for o, s in zip(opts, streams):
                s.wait_event(fork)
                with torch.cuda.stream(s):
                    o.step()
            for s in streams:
                e = torch.cuda.Event()
                e.record(s)
                main_stream.wait_event(e)
        for i in range(6):
            step(i)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        t0 = time.perf_counter()
        ev[0].record()
        for i in range(args.steps):
            step(6 + i)
            ev[i + 1].record()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        per = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps))
        fin = all(bool(torch.isfinite(p).all()) for p in params)
        print(f"lanes {k} {mode}: sizes {[len(x) for x in lanes]} wall mean {dt / args.steps * 1e3:.3f} ms  device median {per[len(per) // 2]:.3f}  min {per[0]:.3f}  "
              f"host enqueue {host / args.steps * 1e3:.3f} ms  loads {[round(x / max(load), 3) for x in load]}  finite {fin}", flush=True)
        del opts, params
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
