"""Feasibility / upper-bound probe for a captured step (review item 3; NOT product code): capture ONE KWNS4.step() of the GPT-2-small plan into a
HIP graph through torch.cuda.graph and replay it.  The replay re-uses the captured step's scalars (Philox offset, balance list), so its RESULTS are
not a valid trajectory; its TIME is what a properly parameterised captured step could reach: wall-clock mean of 20 replays after a synchronize, against
the eager loop's mean and device median on the same box.  Also answers whether the cooperative norm-bound launch can be captured at all."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import psgd_torch_amd
dev = "cuda:0"
shapes = bench.gpt2_shapes()
gen = torch.Generator(device=dev).manual_seed(1)
params = [torch.nn.Parameter(0.02 * torch.randn(*s, device=dev, generator=gen)) for s in shapes]
opt = psgd_torch_amd.KWNS4(params, preconditioner_dtype=torch.bfloat16)
for p in params:
    p.grad = 0.01 * torch.randn(p.shape, device=dev, generator=gen)
for i in range(8):
    opt.step()
torch.cuda.synchronize()


def timed(fn, n=20):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(n):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    per = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    return wall, per[n // 2], per[0]


for rep in range(2):
    w, med, mn = timed(opt.step)
    print(f"eager   : wall mean {w:.4f} ms  device median {med:.4f}  min {mn:.4f}", flush=True)
try:
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(3):
            opt.step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        opt.step()
    torch.cuda.synchronize()
    for rep in range(2):
        w, med, mn = timed(g.replay)
        print(f"captured: wall mean {w:.4f} ms  device median {med:.4f}  min {mn:.4f}", flush=True)
    ok = all(bool(torch.isfinite(p.data).all()) for p in params)
    print("parameters finite after the replays:", ok)
except Exception as e:      # noqa: BLE001
    print("capture failed:", type(e).__name__, str(e)[:600], flush=True)
