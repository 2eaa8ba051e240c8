import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from psgd_torch_amd import _lib
lib = _lib.lib(); dev = "cuda:0"
M, N, K, flag = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dt = torch.bfloat16
A = torch.randn(1, M, K, device=dev).to(dt); B = torch.randn(1, N, K, device=dev).to(dt); Cc = torch.empty(1, M, N, device=dev, dtype=dt)
ms = C.c_float()
_lib.check(lib.psgdk_test_gemm_bench(A.data_ptr(), B.data_ptr(), Cc.data_ptr(), None, 0, M, N, K, 1, flag, 10, C.byref(ms), _lib.current_stream()))
print(M, N, K, flag, ms.value * 1e3, "us")
