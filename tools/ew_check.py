"""Hashes of the engine's whole state and work arenas and of the parameters after three KWNS4 steps on GPT-2-small shapes: two builds (or
one build under two debug settings) that print the same three digests did the same arithmetic, bit for bit."""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import psgd_torch_amd
dev = "cuda:0"
shapes = bench.gpt2_shapes()
gen = torch.Generator(device=dev).manual_seed(1)
params = [torch.nn.Parameter(0.02 * torch.randn(*s, device=dev, generator=gen)) for s in shapes]
opt = psgd_torch_amd.KWNS4(params, preconditioner_dtype=torch.bfloat16 if len(sys.argv) < 2 or sys.argv[1] != "fp32" else torch.float32)
for i in range(3):
    for p in params:
        p.grad = 0.01 * torch.randn(p.shape, device=dev, generator=gen)
    opt.step()
torch.cuda.synchronize()
eng = next(iter(opt._buckets.values())).engine
h = lambda t: hashlib.sha256(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()[:16]
hp = hashlib.sha256()
for p in params:
    hp.update(p.detach().cpu().numpy().tobytes())
print("state", h(eng.state_arena), "work", h(eng.work_arena), "params", hp.hexdigest()[:16])
