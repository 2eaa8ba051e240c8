"""GPU parity tests of the HIP LRA path (-m gpu), through the C ABI (psgd_torch_amd.lra -> libpsgdk.so).

Tolerances (relative Frobenius vs the reference's own output on identical replayed draws): fp32 <= 5e-5 per update
(U, V, d, L, h); bf16: error vs the fp64 oracle trajectory <= 1.5 x the reference-bf16's own error + 3e-2."""
import numpy as np
import pytest
import torch

from helpers import DT, T, golden_names, load, relerr
from oracle import psgd_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("name", golden_names("lra_") + golden_names("lrabig_"))      # lrabig_: rank 32 (two threads per row)
def test_lra_functional_vs_golden(name):
    from psgd_torch_amd import lra
    z = load(name)
    Tn = int(z["T"])
    for dn in [d for d in ("fp32", "bf16") if f"{d}_t0_h" in z.files]:
        dt = DT[dn]
        UVd = [T(z["U0"], dt).to(DEV).contiguous(), T(z["V0"], dt).to(DEV).contiguous(), T(z["d0"], dt).to(DEV).contiguous()]
        Luvd = [torch.zeros([], dtype=torch.float32, device=DEV) for _ in range(3)]
        U64 = [T(z["U0"], torch.float64), T(z["V0"], torch.float64), T(z["d0"], torch.float64)]
        L64 = [torch.zeros([], dtype=torch.float64) for _ in range(3)]
        for t in range(Tn):
            g = T(z[f"g{t}"], dt)
            vn = T(z[f"{dn}_t{t}_vnoise"], dt)
            coin = float(z[f"{dn}_t{t}_coin"])
            lra.update_precond_lra_whiten(UVd, Luvd, g.to(DEV), lr=float(z["lr"]), betaL=float(z["betaL"]), damping=float(z["damping"]),
                                          v_noise=vn.to(DEV), coin=coin)
            h = lra.precond_grad_lra(UVd, g.to(DEV))
            orc.update_precond_lra_whiten(U64, L64, g.double(), vn.double(), coin, lr=float(z["lr"]), betaL=float(z["betaL"]),
                                          damping=float(z["damping"]))
            h64 = orc.precond_grad_lra(U64, g.double())
            checks = [("h", h, z[f"{dn}_t{t}_h"], h64)]
            for k, nm in enumerate(("U", "V", "d")):
                checks.append((nm, UVd[k], z[f"{dn}_t{t}_{nm}"], U64[k]))
            for k, nm in enumerate(("Lu", "Lv", "Ld")):
                checks.append((nm, Luvd[k], z[f"{dn}_t{t}_{nm}"], L64[k]))
            for what, got, gold, truth in checks:
                if dn == "fp32":
                    assert relerr(got, gold) <= 5e-5 * (t + 1), (name, dn, t, what, relerr(got, gold))
                else:
                    e_hip, e_ref = relerr(got, truth), relerr(gold, truth)
                    assert e_hip <= 1.5 * e_ref + 3e-2, (name, dn, t, what, e_hip, e_ref)


@pytest.mark.parametrize("name", golden_names("lrawhiten_"))
def test_lrawhiten_step_vs_golden(name):
    from psgd_torch_amd import lra
    z = load(name)
    Tn = int(z["T"])
    kw = {}
    for k in z.files:
        if k.startswith("kw_"):
            v = z[k]
            nm = k[3:]
            if v.dtype == np.bool_:
                kw[nm] = bool(v)
            elif nm == "rank_of_approximation":
                kw[nm] = int(v)
            else:
                kw[nm] = None if np.isnan(float(v)) else float(v)
    params = [torch.nn.Parameter(T(z[f"p{i}_init"], torch.float32).to(DEV)) for i in range(3)]
    opt = lra.LRAWhiten(params, **kw)
    opt._UVd[0].copy_(T(z["U0"], torch.float32))       # the reference's own random init, replayed
    opt._UVd[1].copy_(T(z["V0"], torch.float32))
    for t in range(Tn):
        cs = [T(z[f"t{t}_g{i}"], torch.float32).to(DEV) for i in range(3)]
        nd = int(z[f"t{t}_ndraws"])
        draws = [z[f"t{t}_draw{k}"] for k in range(nd)]
        u = iter([float(draws[0])] + ([float(draws[2])] if nd > 1 else []))
        opt._uniform = lambda: next(u)
        opt._v_noise = (lambda d=draws: T(d[1], torch.float32).to(DEV)) if nd > 1 else None

        def closure():
            return sum((p * c).sum() for p, c in zip(params, cs))
        opt.step(closure)
        for i in range(3):
            assert relerr(params[i].data, z[f"t{t}_p{i}"]) <= 2e-6 * (t + 1), (name, t, i)
        for k, nm in enumerate(("U", "V", "d")):
            assert relerr(opt._UVd[k], z[f"t{t}_{nm}"]) <= 5e-5 * (t + 1), (name, t, nm)


def test_lra_known_answer():
    """misc/psgd_lra_verification.py restated: H = diag + low rank; after annealed updates precond_grad(g) ~ v."""
    from psgd_torch_amd import lra
    torch.manual_seed(0)
    N, r = 64, 5
    gen = torch.Generator().manual_seed(1)
    Uh = torch.randn(N, 2, generator=gen) / N ** 0.5
    H = torch.diag(0.5 + torch.rand(N, generator=gen)) + Uh @ Uh.t()
    Hd = H.to(DEV)
    U = torch.randn(N, r, generator=gen); U *= 0.1 ** 0.5 / torch.linalg.vector_norm(U)
    V = torch.randn(N, r, generator=gen); V *= 0.1 ** 0.5 / torch.linalg.vector_norm(V)
    UVd = [U.to(DEV).contiguous(), V.to(DEV).contiguous(), torch.ones(N, 1, device=DEV)]
    Luvd = [torch.zeros([], device=DEV) for _ in range(3)]
    dgen = torch.Generator(device=DEV).manual_seed(2)
    iters = 4000
    for it in range(iters):
        v = torch.randn(N, 1, device=DEV, generator=dgen)
        g = Hd @ v
        lra.update_precond_lra_whiten(UVd, Luvd, g, lr=0.1 * (1 - it / iters) + 0.01, betaL=0.9, damping=0.0)
    h = lra.precond_grad_lra(UVd, g)
    assert relerr(h, v) < 0.15, relerr(h, v)


@pytest.mark.parametrize("dn", ["fp32", "bf16"])
@pytest.mark.parametrize("r", [10, 24, 40])
def test_gram_recurrence_tracks_the_true_grams(dn, r, monkeypatch):
    """Round 6: the Grams of psgd.py:1006 carried from update to update (lra_gram_recur_kernel) instead of read from the factors.  The same
    sequence of updates (both branches of the U-or-V coin, replayed noise) with the recurrence on (re-read every 16 updates: never, here) and
    off (psgd.py:1006 as written, every update): the factors, d and the Lipschitz estimates must agree to rounding -- fp32: 2e-5 after 8
    updates; bf16: the two runs round differently, the bound is the golden tests' own (a few bf16 ulp of the factors)."""
    from psgd_torch_amd import lra
    dt = DT[dn]
    N = 300_001
    gen = torch.Generator().manual_seed(r)
    U0 = torch.randn(N, r, generator=gen); U0 *= 0.1 ** 0.5 / torch.linalg.vector_norm(U0)
    V0 = torch.randn(N, r, generator=gen); V0 *= 0.1 ** 0.5 / torch.linalg.vector_norm(V0)
    gs = [torch.randn(N, 1, generator=gen) * torch.linspace(0.2, 3.0, N).reshape(N, 1) for _ in range(8)]
    vs = [torch.randn(N, 1, generator=gen) for _ in range(8)]
    coins = [0.1, 0.9, 0.9, 0.1, 0.1, 0.9, 0.1, 0.9]
    runs = []
    for every in (16, 0):
        monkeypatch.setattr(lra, "GRAM_EVERY", every)
        UVd = [U0.to(dt).to(DEV).contiguous(), V0.to(dt).to(DEV).contiguous(), torch.ones(N, 1, dtype=dt, device=DEV)]
        Luvd = [torch.zeros([], dtype=torch.float32, device=DEV) for _ in range(3)]
        for g, v, c in zip(gs, vs, coins):
            lra.update_precond_lra_whiten(UVd, Luvd, g.to(dt).to(DEV), lr=0.2, betaL=0.9, damping=1e-9, v_noise=v.to(dt).to(DEV), coin=c)
        assert UVd[2]._psgdk_lra.gram_every == every
        h = lra.precond_grad_lra(UVd, gs[0].to(dt).to(DEV))
        torch.cuda.synchronize()
        runs.append(([x.float().cpu() for x in UVd] + [h.float().cpu()], [float(x) for x in Luvd]))
    (a, la), (b, lb) = runs
    tol = 2e-5 if dn == "fp32" else 2e-2
    for nm, x, y in zip(("U", "V", "d", "h"), a, b):
        assert torch.isfinite(x).all()
        assert relerr(x, y) <= tol, (nm, relerr(x, y))
    for x, y in zip(la, lb):
        assert abs(x - y) <= tol * abs(y), (la, lb)
    # and the factors did move: the rotation + eight rank-1 steps are not a no-op
    assert relerr(a[0], U0) > 1e-3


def test_gram_recurrence_notices_a_factor_written_by_somebody_else():
    """The engine cannot see a caller overwriting U between two updates; torch's version counter can: the Python host then has the Grams
    re-read (psgdk_lra_state_changed).  Without that the carried Grams would describe the OLD factor."""
    from psgd_torch_amd import lra
    N, r = 100_000, 10
    gen = torch.Generator().manual_seed(1)
    mk = lambda: (torch.randn(N, r, generator=gen) * (0.1 ** 0.5 / (N * r) ** 0.5)).to(DEV)
    g = torch.randn(N, 1, generator=gen).to(DEV)
    v = torch.randn(N, 1, generator=gen).to(DEV)
    U1, V1, U2 = mk(), mk(), mk()
    res = []
    for poke in (True, False):
        UVd = [U1.clone(), V1.clone(), torch.ones(N, 1, device=DEV)]
        Luvd = [torch.zeros([], device=DEV) for _ in range(3)]
        if not poke:
            UVd[0].copy_(3.0 * U2)                   # the reference run starts from the poked factor and reads its Grams as a first update does
        else:
            lra.update_precond_lra_whiten(UVd, Luvd, g, v_noise=v, coin=0.1)      # Grams now carried ...
            UVd[0].copy_(3.0 * U2); UVd[1].copy_(V1); UVd[2].fill_(1.0)           # ... and the caller replaces the state
            for x in Luvd:
                x.zero_()
        lra.update_precond_lra_whiten(UVd, Luvd, g, v_noise=v, coin=0.9)
        torch.cuda.synchronize()
        res.append([x.cpu() for x in UVd])
    for x, y in zip(*res):
        assert relerr(x, y) <= 1e-6, relerr(x, y)
