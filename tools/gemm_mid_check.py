"""The 256 x 128 tiling (gemm_nt_mid_kernel) against the 128 x 128 kernel (bitwise: same MFMA, K in the same order) and against torch's
fp64 product, through psgdk_test_gemm_nt; then A/B timings of the in-step shapes through psgdk_test_gemm_bench (one process, interleaved)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from psgd_torch_amd import _lib
lib = _lib.lib()
dev = "cuda:0"
BIG, MID, KS, LATE, NOEPI = 1024, 1 << 24, 1 << 25, 1 << 26, 256
bad = 0
print("# correctness: mid tiling vs 128 x 128 (bitwise) and vs fp64")
for (M, N, K) in ((256, 128, 64), (512, 768, 128), (768, 768, 768), (2304, 768, 768), (320, 192, 1024), (64, 64, 64), (192, 320, 192), (832, 64, 448)):
    for dt, tol in ((torch.bfloat16, 2e-2), (torch.float32, 2e-6)):
        for mode in ("C", "T", "CT", "S"):
            if mode == "S" and M != N:
                continue
            A = torch.randn(M, K, device=dev).to(dt)
            B = A if mode == "S" else torch.randn(N, K, device=dev).to(dt)
            outs = {}
            for name, fl in (("128", 0), ("mid", MID), ("mid-late", MID | LATE)):
                Cc = torch.zeros(M, N, device=dev, dtype=dt); Ct = torch.zeros(N, M, device=dev, dtype=dt)
                _lib.check(lib.psgdk_test_gemm_nt(A.data_ptr(), B.data_ptr(), Cc.data_ptr() if mode != "T" else None,
                                                  Ct.data_ptr() if mode in ("T", "CT") else None, _lib.dtype_code(dt), M, N, K, K, K, N, M,
                                                  fl | (1 if mode == "S" else 0), _lib.current_stream()))
                outs[name] = (Cc, Ct)
            ref = A.double() @ B.double().t()
            Cm, Ctm = outs["mid"]
            e = 0.0
            if mode != "T":
                e = max(e, float((Cm.double() - ref).norm() / ref.norm()))
            if mode in ("T", "CT"):
                e = max(e, float((Ctm.double().t() - ref).norm() / ref.norm()))
            same = all(torch.equal(outs["128"][i], outs[k][i]) for k in ("mid", "mid-late") for i in (0, 1))
            ok = e < tol and same
            bad += not ok
            print(f"  {M}x{N}x{K} {str(dt)[6:]:8s} {mode:2s} err {e:.2e} bitwise == 128x128: {same}  {'OK' if ok else 'FAIL'}", flush=True)
print("# FAILURES:", bad)


def run(M, N, K, batch, flags, mode="C", iters=10, dt=torch.bfloat16):
    A = torch.randn(batch, M, K, device=dev).to(dt); B = torch.randn(batch, N, K, device=dev).to(dt)
    Cc = torch.empty(batch, M, N, device=dev, dtype=dt); Ct = torch.empty(batch, N, M, device=dev, dtype=dt)
    ms = C.c_float()
    _lib.check(lib.psgdk_test_gemm_bench(A.data_ptr(), B.data_ptr(), Cc.data_ptr() if "C" in mode else None,
                                         Ct.data_ptr() if "T" in mode else None, _lib.dtype_code(dt), M, N, K, batch, flags, iters,
                                         C.byref(ms), _lib.current_stream()))
    return ms.value * 1e3


def ab(label, M, N, K, batch=1, mode="C", extra=0, rounds=3, dt=torch.bfloat16, sym=0):
    variants = (("128", 0), ("mid", MID), ("pipe", BIG))
    res = {k: [] for k, _ in variants}
    for _ in range(rounds):
        for k, f in variants:
            res[k].append(run(M, N, K, batch, f | extra | sym, mode=mode, dt=dt))
    fl = 2.0 * M * N * K * batch * (0.5 * (1 + 64 / M) if sym else 1.0)
    out = "  ".join(f"{k}: {min(v):7.1f} us {fl / min(v) / 1e6:6.0f} TF" for k, v in res.items())
    print(f"{label:38s} M={M:6d} N={N:5d} K={K:5d} b={batch:2d} {mode:2s} | {out}", flush=True)


print("# in-step shapes (plain epilogues; the fused ones are timed by tools/stage_bench.py)")
ab("62 x 768^3, C", 768, 768, 768, batch=62, mode="C")
ab("  no epilogue", 768, 768, 768, batch=62, mode="C", extra=NOEPI)
ab("62 x 768^3, C + Ct", 768, 768, 768, batch=62, mode="CT")
ab("62 x 768^3 symmetric (P = Q^T Q)", 768, 768, 768, batch=62, mode="C", sym=1)
ab("mode Grams K=3072 x 24 sym", 768, 768, 3072, batch=24, mode="C", sym=1)
ab("mode Grams K=2304 x 12 sym", 768, 768, 2304, batch=12, mode="C", sym=1)
ab("X P, transposed out (update)", 131072, 768, 768, mode="T")
ab("123 x 1024^3, C", 1024, 1024, 1024, batch=123, mode="C")
ab("123 x 1024^3, C + Ct", 1024, 1024, 1024, batch=123, mode="CT")
ab("fp32 24 x 320^3", 320, 320, 320, batch=24, dt=torch.float32)
