"""Times the four-wave 256 x 256 kernel's variants against the eight-wave one on long-K and in-step shapes.  python tools/w4_bench.py [lib.so]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from psgd_torch_amd import _lib
if len(sys.argv) > 1 and sys.argv[1].endswith(".so"):
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
lib = _lib.lib()
dev = "cuda:0"
BIG, W4, NOEPI = 1024, 1 << 26, 256


def run(M, N, K, batch, flags, label, iters=10, tmajor=False):
    dt = torch.bfloat16
    A = torch.randn(batch, M, K, device=dev).to(dt); B = torch.randn(batch, N, K, device=dev).to(dt)
    Cc = torch.empty(batch, M, N, device=dev, dtype=dt)
    ms = C.c_float()
    rc = lib.psgdk_test_gemm_bench(A.data_ptr(), B.data_ptr(), None if tmajor else Cc.data_ptr(), Cc.data_ptr() if tmajor else None, 0, M, N, K, batch,
                                   flags, iters, C.byref(ms), _lib.current_stream())
    if rc:
        print(f"{label:44s} rc {rc}", flush=True)
        return
    print(f"{label:44s} {M:6d} x {N:5d} x {K:5d} x{batch:3d}: {ms.value * 1e3:8.1f} us  {2.0 * M * N * K * batch / ms.value / 1e9:7.1f} TF/s", flush=True)


# (the interleave variants of round 5 -- 2, 3 or 8 matrix instructions between two DMA pieces -- measured within 2 % of the one that ships, 4)
variants = [("pipe (8 waves)", BIG), ("w4", BIG | W4)]
for rnd in range(2):
    for name, fl in variants:
        run(4096, 4096, 4096, 1, fl | NOEPI, f"{name}: 4096^3 no epilogue")
        run(4096, 4096, 4096, 1, fl, f"{name}: 4096^3")
        run(8192, 8192, 8192, 1, fl | NOEPI, f"{name}: 8192^3 no epilogue", iters=4)
        run(131072, 768, 768, 1, fl | NOEPI, f"{name}: in-step X P no epilogue")
        run(131072, 768, 768, 1, fl, f"{name}: in-step X P")
        run(131072, 768, 768, 1, fl, f"{name}: in-step X P t-major", tmajor=True)
        run(768, 768, 768, 62, fl | NOEPI, f"{name}: 62 x 768^3 no epilogue")
        run(768, 768, 768, 62, fl, f"{name}: 62 x 768^3")
run(768, 768, 768, 62, 0, "128x128: 62 x 768^3")
run(768, 768, 768, 62, NOEPI, "128x128: 62 x 768^3 no epilogue")
