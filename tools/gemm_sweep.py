"""GEMM kernel micro-sweep on the GPU (hipEvent-timed, same kernel the engine uses)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from psgd_torch_amd import _lib
lib = _lib.lib()
dev = "cuda:0"
def run(M, N, K, batch=1, mode="C", sym=0, dt=torch.bfloat16, iters=20):
    A = torch.randn(batch, M, K, device=dev).to(dt); B = torch.randn(batch, N, K, device=dev).to(dt)
    Cc = torch.empty(batch, M, N, device=dev, dtype=dt); Ct = torch.empty(batch, N, M, device=dev, dtype=dt)
    ms = C.c_float()
    _lib.check(lib.psgdk_test_gemm_bench(A.data_ptr(), B.data_ptr(), Cc.data_ptr() if "C" in mode else None,
                                         Ct.data_ptr() if "T" in mode else None, _lib.dtype_code(dt), M, N, K, batch, sym, iters,
                                         C.byref(ms), _lib.current_stream()))
    fl = 2.0 * M * N * K * batch * (0.583 if sym == 1 else 1.0)
    tiles = ((M + 127) // 128) * ((N + 127) // 128) * batch
    print(f"M={M:6d} N={N:5d} K={K:5d} b={batch:3d} mode={mode:2s} sym={sym} {str(dt)[6:]:8s}: {ms.value*1e3:8.1f} us  {fl/ms.value/1e9:7.1f} TF  tiles={tiles} us/tile-round={ms.value*1e3/max(1,tiles/512):.1f}")
for K in (64, 256, 768, 3072):
    run(16384, 768, K)
for mode in ("C", "T", "CT"):
    run(16384, 768, 768, mode=mode)
run(768, 768, 768, batch=62); run(768, 768, 768, batch=62, mode="CT"); run(768, 768, 768, batch=62, sym=1)
run(768, 768, 3072, batch=48, sym=1)
run(4096, 4096, 4096); run(8192, 8192, 8192, iters=5)
run(16384, 768, 768, dt=torch.float32); run(4096, 4096, 4096, dt=torch.float32, iters=5)
print("--- big tiling (256x256, 8 waves) vs small")
for shape in ((16384, 768, 768, 1), (50304, 768, 768, 1), (16384, 768, 3072, 1), (768, 768, 768, 62), (1024, 1024, 1024, 48), (4096, 4096, 4096, 1), (8192, 8192, 8192, 1)):
    M, N, K, b = shape
    for flag in (0, 1024):
        run(M, N, K, batch=b, sym=flag, iters=5 if M * N * K > 1e11 else 20)
        run(M, N, K, batch=b, sym=flag | 256, iters=5 if M * N * K > 1e11 else 20)
run(768, 768, 768, batch=62, mode="CT", sym=1024); run(768, 768, 768, batch=62, sym=1025); run(768, 768, 3072, batch=48, sym=1025)
run(16384, 768, 768, mode="T", sym=0); run(16384, 768, 768, mode="T", sym=1024); run(768, 768, 768, batch=62, mode="CT", sym=0)
