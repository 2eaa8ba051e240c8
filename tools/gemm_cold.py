"""Is the in-step slowdown of the skinny X P products (N = K = 768) a cold-operand effect?  Same kernel, same tile count per
launch, operands rotated over several buffer sets so that a launch finds nothing of its A in the 256 MB Infinity Cache."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from psgd_torch_amd import _lib
lib = _lib.lib()
dev = "cuda:0"
BIG, NOEPI, NOMAIN = 1024, 256, 512


def timed(fn, iters):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    fn(0)
    torch.cuda.synchronize()
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def case(M, nsets, flags, mode, label):
    dt = torch.bfloat16
    N = K = 768
    As = [torch.randn(M, K, device=dev).to(dt) for _ in range(nsets)]
    B = torch.randn(N, K, device=dev).to(dt)
    Cs = [torch.empty(M, N, device=dev, dtype=dt) for _ in range(nsets)]
    st = _lib.current_stream()

    def fn(i):
        A, Cc = As[i % nsets], Cs[i % nsets]
        _lib.check(lib.psgdk_test_gemm_launch(A.data_ptr(), B.data_ptr(), Cc.data_ptr() if mode == "C" else None,
                                              Cc.data_ptr() if mode == "T" else None, 0, M, N, K, flags, st))
    us = timed(fn, 12)
    print(f"{label:44s} M={M:6d} sets={nsets} mode={mode}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.0f} TF   ({M * K * 2 * 2 / us / 1e6:.2f} TB/s in+out)", flush=True)


NOSTORE = 4096
for mode in ("T", "C"):
    for nsets in (1, 6):
        for rep in range(2):
            case(161792, nsets, BIG, mode, "pipe persistent")
            case(161792, nsets, BIG | NOSTORE, mode, "pipe persistent, epilogue without stores")
            case(161792, nsets, BIG | NOEPI, mode, "pipe persistent, no epilogue")
            case(161792, nsets, BIG | 16384, mode, "pipe 1 wg/tile")
            case(161792, nsets, BIG | 16384 | NOSTORE, mode, "pipe 1 wg/tile, epilogue without stores")
            case(161792, nsets, BIG | 16384 | NOEPI, mode, "pipe 1 wg/tile, no epilogue")
            case(161792, nsets, BIG | NOMAIN, mode, "pipe persistent, no main loop (stores only)")
