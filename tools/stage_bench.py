"""The grouped-GEMM stages of a real GPT-2-small plan, each launched back to back in isolation (same problems, tile tables and
buffers as in a step), with the alternative main loops / launch shapes -- to separate what a stage costs from what the step's
cache state costs it."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import psgd_torch_amd
from psgd_torch_amd import _lib
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
eng = next(iter(opt._buckets.values())).engine
lib = _lib.lib()
names = ["upd_a (X P -> Pg^T)", "app_a (ema P -> h)", "gram", "qupd", "rq", "rrq", "P = Q^T Q", "upd_b", "app_b"]
vnames = {0: "as bound", 2: "1 wg/tile", 3: "128x128", 4: "256x256", 5: "no sums", 6: "no sums/scale", 7: "no stores", 8: "no epilogue", 14: "128x128 (forced)", 20: "8-wave", 21: "4-wave", 22: "8-wave no epi", 23: "4-wave no epi", 24: "8-wave no st", 25: "4-wave no st", 26: "8-wave reg st", 27: "4-wave reg st", 56: "reg stores", 60: "ring", 61: "ring no epi", 30: "ping-pong", 31: "ping-pong no epi", 32: "ping-pong no epi/DMA"}
for rnd in range(2):
    for which, nm in list(enumerate(names))[:9]:
        out = []
        for v in ((0, 5, 6, 7, 8) if len(sys.argv) < 3 else tuple(int(x) for x in sys.argv[2].split(','))):
            ms = C.c_float()
            rc = lib.psgdk_test_stage_bench(eng._plan, which, v, 10, C.byref(ms), _lib.current_stream())
            out.append(f"{vnames[v]} {ms.value * 1e3:7.1f}" if rc == 0 else f"{vnames[v]} rc={rc}")
        print(f"{nm:24s} | " + " | ".join(out), flush=True)
