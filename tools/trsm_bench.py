"""The two right triangular solves X = Y inv(U) (EQ geometry, psgd.py:288-293) alone, on GPT-2-small's row count: the fp32-core
kernel vs the bf16 LDS-panel kernel, and the latter with parts switched off.  Run on the GPU box."""
import ctypes as C
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))

from psgd_torch_amd import _lib

lib = _lib.lib()
dev = "cuda:0"
rows_list = [int(x) for x in sys.argv[1:]] or [161920, 16384]
for d in (768, 1024):
    for rows in rows_list:
        g = torch.Generator().manual_seed(0)
        U = (torch.triu(torch.randn(d, d, generator=g) / d ** 0.5) + 1.5 * torch.eye(d)).to(torch.bfloat16).to(dev)
        Ut = U.t().contiguous()
        Y = torch.randn(rows, d, generator=g).to(torch.bfloat16).to(dev)
        Ot = torch.zeros(d, rows, dtype=torch.bfloat16, device=dev)
        On = torch.zeros(rows, d, dtype=torch.bfloat16, device=dev)
        ms = C.c_float()
        st = _lib.current_stream()
        res = {}
        for name, ut, dbg, nat in [("fp32 cores", None, 0, False), ("bf16 panel", Ut, 0, False), ("  + natural output too", Ut, 0, True),
                                   ("  no update loop", Ut, 1, False), ("  no diagonal step", Ut, 2, False), ("  no stores", Ut, 4, False),
                                   ("  load + barriers only", Ut, 7, False), ("64-row panels", Ut, 8, False),
                                   ("  64: load + barriers only", Ut, 15, False)]:
            _lib.check(lib.psgdk_test_trsm_bench(Y.data_ptr(), U.data_ptr(), ut.data_ptr() if ut is not None else None,
                                                 On.data_ptr() if nat else None, Ot.data_ptr(), rows, d, 5, dbg, C.byref(ms), None, st))
            res[name] = ms.value * 1e3
        gf = rows * d * d / 1e9
        print(f"d={d} rows={rows}: " + " | ".join(f"{k} {v:.1f} us" for k, v in res.items()) + f" | {gf / res['bf16 panel'] * 1e3:.0f} TF/s")
        # phase stamps of the first panel (shader clocks): start, Y in LDS, then per block: barrier | update loop | A operand staged |
        # diagonal step done; then panel finished, outputs stored
        stamps = torch.zeros(256, dtype=torch.int64, device=dev)
        _lib.check(lib.psgdk_test_trsm_bench(Y.data_ptr(), U.data_ptr(), Ut.data_ptr(), None, Ot.data_ptr(), rows, d, 1, 16, C.byref(ms),
                                             stamps.data_ptr(), st))
        torch.cuda.synchronize()
        t = stamps.cpu().tolist()
        nb = (d + 63) // 64
        n = 2 + 4 * nb + 2
        dlt = [t[i + 1] - t[i] for i in range(n - 1)]
        print(f"   stamps d={d}: Y load {dlt[0]} | " + " ".join(f"[b{j}: wait {dlt[1 + 4 * j]} upd {dlt[2 + 4 * j]} stage {dlt[3 + 4 * j]} diag {dlt[4 + 4 * j]}]"
                                                                for j in range(nb)) + f" | tail {dlt[1 + 4 * nb:]}  total {t[n - 1] - t[0]}")
