"""Diagnostics for the two routes of the spectral-norm bound: run-to-run spread of each route and route-vs-route differences
(row sums of squares of the four products, elements of the last block), per chain."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_gpu_nlb import _engine, _run_bound, DEV

for dt, width, nf in ((torch.bfloat16, 768, 62), (torch.float32, 384, 40)):
    eng, _ = _engine(nf, width, dt)
    F, dp = eng.info()["dense_factors"], eng.info()["max_dense_dim"]
    def run(chain, route, seed):
        vsq = torch.zeros(F, 4, 32, device=DEV); v = torch.zeros(F, 2, 32, dp, device=DEV, dtype=dt)
        _run_bound(eng, chain, route, seed, vsq, v)
        torch.cuda.synchronize()
        return vsq, v
    for chain in (0, 1):
        for seed in (11, 12, 13):
            a0, va0 = run(chain, 0, seed); b0, vb0 = run(chain, 0, seed)
            a1, va1 = run(chain, 1, seed); b1, vb1 = run(chain, 1, seed)
            def rel(x, y):
                return ((x - y).abs() / y.abs().clamp_min(1e-30)).amax(dim=(0, 2)).tolist()
            def vdiff(x, y):
                d = (x.float() - y.float()).abs(); n = int((d > 0).sum()); m = float(d.max() / y.float().abs().max())
                return n, m
            print(f"{str(dt)[6:]:8s} chain {chain} seed {seed}: vsq rel diff per product  r0-r0 {['%.1e' % x for x in rel(b0, a0)]}  r1-r1 {['%.1e' % x for x in rel(b1, a1)]}  "
                  f"r1-r0 {['%.1e' % x for x in rel(a1, a0)]} | V4 differing elems/maxrel r0-r0 {vdiff(vb0, va0)} r1-r1 {vdiff(vb1, va1)} r1-r0 {vdiff(va1, va0)}", flush=True)
            # where is the worst row?
            d = ((a1 - a0).abs() / a0.abs().clamp_min(1e-30))
            idx = torch.nonzero(d == d.max())[0].tolist()
            f, p, r = idx
            print(f"    worst at factor {f} product {p} row {r}: r0 {float(a0[f, p, r]):.6e} r1 {float(a1[f, p, r]):.6e}; row sums of that factor/product r0 {a0[f, p, :4].tolist()}")
