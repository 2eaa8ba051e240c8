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
