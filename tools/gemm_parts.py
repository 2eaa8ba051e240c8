"""Splits the 256x256-tile GEMM's time on the in-step shape (X P: M = all rows of the step, N = K = 768) into main loop,
epilogue and fixed cost, with the kernel's own debug flags (256 = no epilogue, 512 = no main loop)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from psgd_torch_amd import _lib
lib = _lib.lib()
dev = "cuda:0"


def run(M, N, K, batch, flags, mode="C", iters=20, label=""):
    dt = torch.bfloat16
    A = torch.randn(batch, M, K, device=dev).to(dt); B = torch.randn(batch, N, K, device=dev).to(dt)
    Cc = torch.empty(batch, M, N, device=dev, dtype=dt); Ct = torch.empty(batch, N, M, device=dev, dtype=dt)
    ms = C.c_float()
    _lib.check(lib.psgdk_test_gemm_bench(A.data_ptr(), B.data_ptr(), Cc.data_ptr() if "C" in mode else None,
                                         Ct.data_ptr() if "T" in mode else None, _lib.dtype_code(dt), M, N, K, batch, flags, iters,
                                         C.byref(ms), _lib.current_stream()))
    bm = bn = 256 if flags & 1024 else 128
    tiles = ((M + bm - 1) // bm) * ((N + bn - 1) // bn) * batch
    slots = 256 if flags & 1024 else 512
    rounds = -(-tiles // slots)
    print(f"{label:34s} M={M:6d} N={N:4d} K={K:4d} b={batch:2d} mode={mode:2s}: {ms.value*1e3:7.1f} us  {2.0*M*N*K*batch/ms.value/1e9:7.1f} TF  "
          f"tiles={tiles} rounds={tiles/slots:.2f} us/round={ms.value*1e3/rounds:.1f}", flush=True)


for M in (65536 * 2,):          # 1899 tiles = 7.4 rounds; 1536 tiles = exactly 6 rounds
    for big in (1024, 0):
        tag = "256x256" if big else "128x128"
        run(M, 768, 768, 1, big, label=tag + " full")
        run(M, 768, 768, 1, big | 256, label=tag + " no epilogue")
        run(M, 768, 768, 1, big | 4096, label=tag + " epilogue without its stores")
        run(M, 768, 768, 1, big | 512, label=tag + " no main loop")
        run(M, 768, 768, 1, big | 768, label=tag + " neither (launch + tile setup)")
        run(M, 768, 768, 1, big | 512 | 4096, label=tag + " epilogue alone, no stores")
for big in (1024, 0):
    tag = "256x256" if big else "128x128"
    run(768, 768, 768, 62, big, mode="CT", label=tag + " 62x768^3 C+Ct full")
    run(768, 768, 768, 62, big | 256, mode="CT", label=tag + " 62x768^3 no epilogue")
    run(768, 768, 768, 62, big | 512, mode="CT", label=tag + " 62x768^3 no main loop")
