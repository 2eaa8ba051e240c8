"""The ping-pong four-wave kernel (gemm_w4p.hiph) against the 128 x 128 tiling: outputs must be bit-identical (same MFMA, same K order,
same rounding points).  python tools/w4p_check.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from psgd_torch_amd import _lib
lib = _lib.lib()
dev = "cuda:0"
st = _lib.current_stream()
BIG, W4P = 1024, 1 << 23
torch.manual_seed(0)
bad = 0
shapes = [(256, 128, 640), (64, 64, 640), (128, 256, 768), (320, 768, 768), (768, 768, 768), (2304, 768, 768), (768, 3072, 1024),
          (50304, 768, 768), (1024, 1024, 1024), (4160, 832, 704), (131072, 768, 768)]
for (M, N, K) in shapes:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    ref_c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    ref_t = torch.zeros(N, M, device=dev, dtype=torch.bfloat16)
    _lib.check(lib.psgdk_test_gemm_nt(A.data_ptr(), B.data_ptr(), ref_c.data_ptr(), None, 0, M, N, K, K, K, N, M, 0, st))
    _lib.check(lib.psgdk_test_gemm_nt(A.data_ptr(), B.data_ptr(), None, ref_t.data_ptr(), 0, M, N, K, K, K, N, M, 0, st))
    if M * N <= 4096 * 4096:
        fp = (A.double() @ B.double().t())
        e0 = ((ref_c.double() - fp).norm() / fp.norm()).item()
    else:
        e0 = float("nan")
    for rep in range(3):
        c = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
        t = torch.full((N, M), float("nan"), device=dev, dtype=torch.bfloat16)
        _lib.check(lib.psgdk_test_gemm_nt(A.data_ptr(), B.data_ptr(), c.data_ptr(), None, 0, M, N, K, K, K, N, M, BIG | W4P, st))
        _lib.check(lib.psgdk_test_gemm_nt(A.data_ptr(), B.data_ptr(), None, t.data_ptr(), 0, M, N, K, K, K, N, M, BIG | W4P, st))
        okc, okt = torch.equal(c, ref_c), torch.equal(t, ref_t)
        if not (okc and okt):
            bad += 1
            nbad_c = int((c != ref_c).sum().item()); nbad_t = int((t != ref_t).sum().item())
            nan_c = int(torch.isnan(c.float()).sum().item()); nan_t = int(torch.isnan(t.float()).sum().item())
            print(f"MISMATCH {M}x{N}x{K} rep {rep}: C equal {okc} ({nbad_c} differ, {nan_c} never written), Ct equal {okt} ({nbad_t} differ, {nan_t} never written)", flush=True)
            if not okc:
                idx = (c != ref_c).nonzero()[:6].tolist()
                print("   first C mismatches (row, col):", idx, flush=True)
            if not okt:
                idx = (t != ref_t).nonzero()[:6].tolist()
                print("   first Ct mismatches (row, col):", idx, flush=True)
    print(f"{M}x{N}x{K}: done (ref relerr vs fp64 {e0:.3e})", flush=True)
print("W4P CHECK", "FAILED" if bad else "OK", bad)
sys.exit(1 if bad else 0)
