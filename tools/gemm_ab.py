"""A/B of the two main loops of the 256x256 tiling (staggered phases vs lock-step) and the 128x128 tiling, interleaved in ONE
process (hipEvent-timed through psgdk_test_gemm_bench), on the shapes of a GPT-2-small / -medium step.  Random bf16 operands."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from psgd_torch_amd import _lib
lib = _lib.lib()
dev = "cuda:0"
BIG, LOCK, NOEPI, NOMAIN = 1024, 2048, 256, 512


def run(M, N, K, batch, flags, mode="C", iters=10, dt=torch.bfloat16):
    A = torch.randn(batch, M, K, device=dev).to(dt); B = torch.randn(batch, N, K, device=dev).to(dt)
    Cc = torch.empty(batch, M, N, device=dev, dtype=dt); Ct = torch.empty(batch, N, M, device=dev, dtype=dt)
    ms = C.c_float()
    _lib.check(lib.psgdk_test_gemm_bench(A.data_ptr(), B.data_ptr(), Cc.data_ptr() if "C" in mode else None,
                                         Ct.data_ptr() if "T" in mode else None, _lib.dtype_code(dt), M, N, K, batch, flags, iters,
                                         C.byref(ms), _lib.current_stream()))
    return ms.value * 1e3


def ab(label, M, N, K, batch=1, mode="C", extra=0, rounds=3, dt=torch.bfloat16, sym=0):
    variants = (("pipe", BIG), ("lock", BIG | LOCK), ("128", 0))
    res = {k: [] for k, _ in variants}
    for _ in range(rounds):
        for k, f in variants:
            res[k].append(run(M, N, K, batch, f | extra | sym, mode=mode, dt=dt))
    fl = 2.0 * M * N * K * batch * (0.5 * (1 + 256 / M) if sym else 1.0)
    out = "  ".join(f"{k}: {min(v):7.1f} us {fl / min(v) / 1e6:6.0f} TF" for k, v in res.items())
    print(f"{label:38s} M={M:6d} N={N:5d} K={K:5d} b={batch:2d} {mode:2s} | {out}", flush=True)


print("# correctness of the staggered main loop vs torch (fp64 reference)")
for (M, N, K, b) in ((256, 256, 64, 1), (512, 768, 128, 1), (768, 768, 768, 3), (2304, 768, 768, 1), (320, 192, 1024, 1), (64, 64, 64, 1)):
    for dt, tol in ((torch.bfloat16, 2e-2), (torch.float32, 2e-6)):
        A = torch.randn(M, K, device=dev).to(dt); B = torch.randn(N, K, device=dev).to(dt)
        Cc = torch.zeros(M, N, device=dev, dtype=dt); Ct = torch.zeros(N, M, device=dev, dtype=dt)
        _lib.check(lib.psgdk_test_gemm_nt(A.data_ptr(), B.data_ptr(), Cc.data_ptr(), Ct.data_ptr(), _lib.dtype_code(dt), M, N, K, K, K, N, M, BIG,
                                          _lib.current_stream()))
        ref = A.double() @ B.double().t()
        e1 = float((Cc.double() - ref).norm() / ref.norm()); e2 = float((Ct.double().t() - ref).norm() / ref.norm())
        Cl = torch.zeros_like(Cc)
        _lib.check(lib.psgdk_test_gemm_nt(A.data_ptr(), B.data_ptr(), Cl.data_ptr(), None, _lib.dtype_code(dt), M, N, K, K, K, N, M, BIG | LOCK,
                                          _lib.current_stream()))
        print(f"  {M}x{N}x{K} {str(dt)[6:]:8s} err C {e1:.2e} Ct {e2:.2e}  bitwise == lock-step: {bool(torch.equal(Cl, Cc))}  {'OK' if max(e1, e2) < tol else 'FAIL'}")

print("# in-step shapes")
ab("X P, transposed out (update)", 131072, 768, 768, mode="T")
ab("X P, normal out (apply)", 131072, 768, 768, mode="C")
ab("  no epilogue", 131072, 768, 768, mode="C", extra=NOEPI)
ab("  no main loop", 131072, 768, 768, mode="C", extra=NOMAIN)
ab("all rows of GPT-2-small", 161920 - 161920 % 256, 768, 768, mode="T")
ab("62 x 768^3, C + Ct", 768, 768, 768, batch=62, mode="CT")
ab("62 x 768^3, C", 768, 768, 768, batch=62, mode="C")
ab("62 x 768^3 symmetric (P = Q^T Q)", 768, 768, 768, batch=62, mode="C", sym=1)
ab("mode Grams K=3072 x 24 sym", 768, 768, 3072, batch=24, mode="C", sym=1)
ab("wte Gram K=50304 (1 problem)", 768, 768, 50304 - 50304 % 64, batch=1, mode="C", sym=1)
print("# GPT-2-medium")
ab("X P medium", 131072, 1024, 1024, mode="T")
ab("123 x 1024^3, C + Ct", 1024, 1024, 1024, batch=123, mode="CT")
print("# large")
ab("4096^3", 4096, 4096, 4096, rounds=2)
ab("8192^3", 8192, 8192, 8192, rounds=2, extra=0)
ab("fp32 16384x768x768", 16384, 768, 768, dt=torch.float32, rounds=2)
