"""The 256 x 256 kernel's main loop on long-K and in-step shapes, against ANOTHER build of the library:  python tools/gemm_parts_lib.py <lib.so>
(timing experiments: variants that drop a wait or the DMA compute garbage -- only the times mean anything)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from psgd_torch_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
lib = _lib.lib()
dev = "cuda:0"


def run(M, N, K, batch, flags, label, iters=10):
    dt = torch.bfloat16
    A = torch.randn(batch, M, K, device=dev).to(dt); B = torch.randn(batch, N, K, device=dev).to(dt)
    Cc = torch.empty(batch, M, N, device=dev, dtype=dt)
    ms = C.c_float()
    _lib.check(lib.psgdk_test_gemm_bench(A.data_ptr(), B.data_ptr(), Cc.data_ptr(), None, _lib.dtype_code(dt), M, N, K, batch, flags, iters,
                                         C.byref(ms), _lib.current_stream()))
    print(f"{os.path.basename(sys.argv[1]):20s} {label:30s} {M:6d} x {N:5d} x {K:5d}: {ms.value * 1e3:8.1f} us  {2.0 * M * N * K * batch / ms.value / 1e9:7.1f} TF/s", flush=True)


which = sys.argv[2] if len(sys.argv) > 2 else "pipe"
if which == "k128":
    run(4096, 4096, 4096, 1, 256, "128x128 4096^3 no epilogue")
    run(768, 768, 768, 62, 0, "128x128 62 x 768^3")
    run(768, 768, 768, 62, 256, "128x128 62 x 768^3 no epilogue")
    sys.exit(0)
run(4096, 4096, 4096, 1, 1024, "4096^3")
run(4096, 4096, 4096, 1, 1024 | 256, "4096^3 no epilogue")
run(8192, 8192, 8192, 1, 1024 | 256, "8192^3 no epilogue")
run(131072, 768, 768, 1, 1024, "in-step X P")
run(131072, 768, 768, 1, 1024 | 256, "in-step X P no epilogue")
