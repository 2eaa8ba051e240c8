"""The two streaming passes of a real GPT-2-small plan (momentum + cast + damped input; clip + parameter update), each launched
back to back in isolation through the engine's own calls, event-timed: GB/s against the passes' algorithmic bytes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import psgd_torch_amd
dev = "cuda:0"
shapes = bench.gpt2_shapes() if len(sys.argv) < 2 or sys.argv[1] != "medium" else bench.gpt2_shapes(n_layer=24, n_embd=1024)
gen = torch.Generator(device=dev).manual_seed(1)
params = [torch.nn.Parameter(0.02 * torch.randn(*s, device=dev, generator=gen)) for s in shapes]
opt = psgd_torch_amd.KWNS4(params, preconditioner_dtype=torch.bfloat16)
for i in range(3):
    for p in params:
        p.grad = 0.01 * torch.randn(p.shape, device=dev, generator=gen)
    opt.step()
torch.cuda.synchronize()
b = next(iter(opt._buckets.values()))
eng = b.engine
own = [b.params[i] for i in b.owned]
grads = [p.grad for p in own]
n = sum(p.numel() for p in own)
pd = [p.data for p in own]


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


for rnd in range(3):
    t_acc = timed(lambda: eng.accumulate(grads, beta=0.9, damp=dict(source=0, damping=1e-9, seed=7, offset=rnd)))
    t_acc0 = timed(lambda: eng.accumulate(grads, beta=0.9))
    eng.precond_grad(0)
    t_emit = timed(lambda: eng.apply_update(pd, 1e-6, 0.0, 2.0, 10.0))
    print(f"accumulate+X {t_acc:7.1f} us ({n * 10 / t_acc / 1e6:5.2f} TB/s) | accumulate alone {t_acc0:7.1f} us ({n * 8 / t_acc0 / 1e6:5.2f} TB/s) | "
          f"apply_update {t_emit:7.1f} us ({n * 10 / t_emit / 1e6:5.2f} TB/s)", flush=True)
