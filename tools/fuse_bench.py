"""The apply of a GPT-2-small plan in isolation, event-timed: two-call route (precond_grad + apply_update) against the fused call
(psgdk_precond_grad_apply), with parts of the fused epilogue's memory traffic switched off (timing experiments: wrong results)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import psgd_torch_amd
dev = "cuda:0"
shapes = bench.gpt2_shapes()
gen = torch.Generator(device=dev).manual_seed(1)
params = [torch.nn.Parameter(0.02 * torch.randn(*s, device=dev, generator=gen)) for s in shapes]
opt = psgd_torch_amd.KWNS4(params, preconditioner_dtype=torch.bfloat16)
for i in range(2):
    for p in params:
        p.grad = 0.01 * torch.randn(p.shape, device=dev, generator=gen)
    opt.step()
torch.cuda.synchronize()
b = next(iter(opt._buckets.values()))
eng = b.engine
pd = [p.data for p in [b.params[i] for i in b.owned]] if hasattr(b, "params") else [p.data for p in params]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def two():
    eng.precond_grad(0)
    eng.apply_update(pd, 1e-6, 0.0, 2.0, 10.0)


for rnd in range(2):
    eng.fuse_update(True, 0)
    t2 = timed(two)
    tg = timed(lambda: eng.precond_grad(0))
    out = [f"two-call {t2:7.1f} us (precond_grad alone {tg:7.1f})"]
    out.append(f"fused {timed(lambda: eng.precond_grad_apply(0, pd, 1e-6, 0.0, 2.0, 10.0)):7.1f}")
    for st in (2000, 4000):
        eng.fuse_update(True, st)
        out.append(f"stagger {st & 0x3fffff}{'all' if st >> 22 else ''} {timed(lambda: eng.precond_grad_apply(0, pd, 1e-6, 0.0, 2.0, 10.0)):7.1f}")
    print(" | ".join(out), flush=True)
