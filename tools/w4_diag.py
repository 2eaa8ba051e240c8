"""Where the four-wave kernel's time goes: epilogue arithmetic vs stores; main loop without DMA / without the barrier (timing only).
python tools/w4_diag.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from psgd_torch_amd import _lib
lib = _lib.lib()
dev = "cuda:0"
BIG, W4, NOEPI, NOSTORE = 1024, 1 << 26, 256, 4096


def run(M, N, K, batch, flags, label, iters=10, tmajor=False):
    dt = torch.bfloat16
    A = torch.randn(batch, M, K, device=dev).to(dt); B = torch.randn(batch, N, K, device=dev).to(dt)
    Cc = torch.empty(batch, M, N, device=dev, dtype=dt)
    ms = C.c_float()
    rc = lib.psgdk_test_gemm_bench(A.data_ptr(), B.data_ptr(), None if tmajor else Cc.data_ptr(), Cc.data_ptr() if tmajor else None, 0, M, N, K, batch,
                                   flags, iters, C.byref(ms), _lib.current_stream())
    if rc:
        print(f"{label:52s} rc {rc}", flush=True)
        return
    print(f"{label:52s} {M:6d} x {N:5d} x {K:5d} x{batch:3d}: {ms.value * 1e3:8.1f} us  {2.0 * M * N * K * batch / ms.value / 1e9:7.1f} TF/s", flush=True)


for rnd in range(int(os.environ.get('W4_DIAG_FIRST', '0'))):
    for name, fl in (("w4", BIG | W4), ("pipe", BIG)):
        for tm in (False, True):
            t = " t-major" if tm else ""
            run(131072, 768, 768, 1, fl, f"{name}: in-step X P{t}", tmajor=tm)
            run(131072, 768, 768, 1, fl | NOEPI, f"{name}: in-step X P{t} no epilogue", tmajor=tm)
            run(131072, 768, 768, 1, fl | NOSTORE, f"{name}: in-step X P{t} no stores", tmajor=tm)
    for var, nm in ((0, "as built"), (4, "no DMA"), (5, "no barrier"), (6, "no DMA, no barrier")):
        fl = BIG | W4 | (var << 27) | NOEPI
        run(4096, 4096, 4096, 1, fl, f"w4 {nm}: 4096^3 no epilogue")
        run(8192, 8192, 8192, 1, fl, f"w4 {nm}: 8192^3 no epilogue", iters=4)
        run(131072, 768, 768, 1, fl, f"w4 {nm}: in-step X P no epilogue")
W4P = 1 << 23
for rnd in range(3):
    for name, fl in (("ping-pong", BIG | W4P), ("w4", BIG | W4), ("pipe", BIG)):
        for tm in (False, True):
            t = " t-major" if tm else ""
            run(131072, 768, 768, 1, fl, f"{name}: in-step X P{t}", tmajor=tm)
            run(131072, 768, 768, 1, fl | NOEPI if name != "ping-pong" else fl | (7 << 27), f"{name}: in-step X P{t} no epilogue", tmajor=tm)
        if name == "ping-pong":
            run(131072, 768, 768, 1, fl | (4 << 27), f"{name}: in-step X P no epilogue, no DMA")
            run(131072, 768, 768, 1, fl | NOSTORE, f"{name}: in-step X P no stores")
            run(4096, 4096, 4096, 1, fl | (7 << 27), f"{name}: 4096^3 no epilogue")
            run(4096, 4096, 4096, 1, fl, f"{name}: 4096^3")
            run(768, 768, 768, 62, fl, f"{name}: 62 x 768^3")
