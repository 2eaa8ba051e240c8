"""The four-wave 256 x 256 kernel (gemm_w4.hiph) against the 128 x 128 tiling: outputs must be bit-identical (same MFMA, same K order,
same epilogue code).  python tools/w4_check.py [lib.so]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from psgd_torch_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
lib = _lib.lib()
dev = "cuda:0"
st = _lib.current_stream()
BIG, W4 = 1024, 1 << 26
torch.manual_seed(0)
bad = 0
shapes = [(256, 256, 128), (64, 64, 128), (256, 256, 192), (320, 768, 768), (768, 768, 768), (2304, 768, 768), (768, 3072, 256),
          (44032, 768, 128), (50304, 768, 768), (1024, 1024, 1024), (4160, 832, 448)]
for (M, N, K) in shapes:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    ref_c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    ref_t = torch.zeros(N, M, device=dev, dtype=torch.bfloat16)
    _lib.check(lib.psgdk_test_gemm_nt(A.data_ptr(), B.data_ptr(), ref_c.data_ptr(), None, 0, M, N, K, K, K, N, M, 0, st))
    _lib.check(lib.psgdk_test_gemm_nt(A.data_ptr(), B.data_ptr(), None, ref_t.data_ptr(), 0, M, N, K, K, K, N, M, 0, st))
    fp = (A.double() @ B.double().t())
    e0 = ((ref_c.double() - fp).norm() / fp.norm()).item()
    for var in range(1):
        for rep in range(2):
            c = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
            t = torch.full((N, M), float("nan"), device=dev, dtype=torch.bfloat16)
            _lib.check(lib.psgdk_test_gemm_nt(A.data_ptr(), B.data_ptr(), c.data_ptr(), None, 0, M, N, K, K, K, N, M, BIG | W4 | (var << 27), st))
            _lib.check(lib.psgdk_test_gemm_nt(A.data_ptr(), B.data_ptr(), None, t.data_ptr(), 0, M, N, K, K, K, N, M, BIG | W4 | (var << 27), st))
            okc, okt = torch.equal(c, ref_c), torch.equal(t, ref_t)
            if not (okc and okt):
                bad += 1
                ec = ((c.double() - fp).norm() / fp.norm()).item()
                et = ((t.double().t() - fp).norm() / fp.norm()).item()
                nbad_c = int((c != ref_c).sum().item()); nbad_t = int((t != ref_t).sum().item())
                print(f"MISMATCH {M}x{N}x{K} var {var} rep {rep}: C equal {okc} ({nbad_c} differ, relerr {ec:.3e}), Ct equal {okt} ({nbad_t} differ, relerr {et:.3e}); ref relerr {e0:.3e}", flush=True)
    print(f"{M}x{N}x{K}: done (ref relerr vs fp64 {e0:.3e})", flush=True)
print("W4 CHECK", "FAILED" if bad else "OK", bad)
sys.exit(1 if bad else 0)
