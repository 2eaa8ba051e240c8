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
    # lrawhiten_bf16_*: bf16 parameters, N = 1710, even rank -- the packed row passes (kernels_lra_pk.hiph) inside the optimizer.  Yardstick as
    # for every bf16 comparison: the fp64 oracle on the same bf16-rounded inputs and recorded draws is the truth, the reference's own bf16
    # result (the golden) sets the scale: HIP error <= 1.5 x its error + one bf16 ulp.
    bf = name.startswith("lrawhiten_bf16")
    dt = torch.bfloat16 if bf else torch.float32
    params = [torch.nn.Parameter(T(z[f"p{i}_init"], dt).to(DEV)) for i in range(3)]
    opt = lra.LRAWhiten(params, **kw)
    opt._UVd[0].copy_(T(z["U0"], torch.float32))       # the reference's own random init, replayed
    opt._UVd[1].copy_(T(z["V0"], torch.float32))
    if bf:
        okw = {k: v for k, v in kw.items() if k != "rank_of_approximation"}
        p64 = [T(z[f"p{i}_init"], torch.float64).clone() for i in range(3)]
        o64 = orc.LRAWhitenOracle(p64, T(z["U0"], torch.float64), T(z["V0"], torch.float64), **okw)
    for t in range(Tn):
        cs = [T(z[f"t{t}_g{i}"], dt).to(DEV) for i in range(3)]
        nd = int(z[f"t{t}_ndraws"])
        draws = [z[f"t{t}_draw{k}"] for k in range(nd)]
        u = iter([float(draws[0])] + ([float(draws[2])] if nd > 1 else []))
        opt._uniform = lambda: next(u)
        opt._v_noise = (lambda d=draws: T(d[1], torch.float32).to(DEV)) if nd > 1 else None

        def closure():
            return sum((p * c).sum() for p, c in zip(params, cs))
        opt.step(closure)
        if not bf:
            for i in range(3):
                assert relerr(params[i].data, z[f"t{t}_p{i}"]) <= 2e-6 * (t + 1), (name, t, i)
            for k, nm in enumerate(("U", "V", "d")):
                assert relerr(opt._UVd[k], z[f"t{t}_{nm}"]) <= 5e-5 * (t + 1), (name, t, nm)
            continue
        o64.step([T(z[f"t{t}_g{i}"], torch.float64) for i in range(3)], float(draws[0]),
                 T(draws[1], torch.float64) if nd > 1 else None, float(draws[2]) if nd > 1 else None)
        if nd > 1:
            assert opt._UVd[2]._psgdk_lra.info()["packed_rows"] == 1536
        for i in range(3):
            e_hip, e_ref = relerr(params[i].data, p64[i]), relerr(z[f"t{t}_p{i}"], p64[i])
            assert e_hip <= 1.5 * e_ref + 7.8125e-3, (name, t, i, e_hip, e_ref)
        for k, nm in enumerate(("U", "V", "d")):
            e_hip, e_ref = relerr(opt._UVd[k], o64.UVd[k]), relerr(z[f"t{t}_{nm}"], o64.UVd[k])
            assert e_hip <= 1.5 * e_ref + 7.8125e-3, (name, t, nm, e_hip, e_ref)


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


def _pk_inputs(N, r, seed):
    gen = torch.Generator().manual_seed(1000 * r + seed)
    bf = torch.bfloat16
    U0 = torch.randn(N, r, generator=gen); U0 *= 0.1 ** 0.5 / torch.linalg.vector_norm(U0)
    V0 = torch.randn(N, r, generator=gen); V0 *= 0.1 ** 0.5 / torch.linalg.vector_norm(V0)
    d0 = 0.5 + torch.rand(N, 1, generator=gen)
    scale = torch.linspace(0.3, 2.5, N).reshape(N, 1)
    gs = [(torch.randn(N, 1, generator=gen) * scale).to(bf) for _ in range(3)]
    vs = [torch.randn(N, 1, generator=gen).to(bf) for _ in range(3)]
    return U0.to(bf), V0.to(bf), d0.to(bf), gs, vs


def _pk_run(U0, V0, d0, gs, vs, coins=(0.2, 0.8, 0.2), offset_by_one_element=False):
    """Three updates (both branches of the U-or-V coin) + an apply after each on the HIP path; returns the host copies and what
    psgdk_lra_info said about the row kernels of the last call."""
    from psgd_torch_amd import lra
    UVd = [U0.to(DEV).contiguous(), V0.to(DEV).contiguous(), d0.to(DEV).contiguous()]
    Luvd = [torch.zeros([], dtype=torch.float32, device=DEV) for _ in range(3)]
    hs, packed = [], []
    for g, v, c in zip(gs, vs, coins):
        gd = g.to(DEV)
        if offset_by_one_element:       # a gradient that starts 2 bytes into its allocation: the 32-bit N-vector accesses cannot be used
            buf = torch.empty(g.numel() + 1, dtype=g.dtype, device=DEV)
            buf[1:].copy_(gd.reshape(-1))
            gd = buf[1:].reshape(g.shape)
            assert gd.data_ptr() % 4 == 2 and gd.is_contiguous()
        lra.update_precond_lra_whiten(UVd, Luvd, gd, lr=0.2, betaL=0.9, damping=1e-9, v_noise=v.to(DEV), coin=c)
        packed.append(UVd[2]._psgdk_lra.info()["packed_rows"])
        hs.append(lra.precond_grad_lra(UVd, gd).float().cpu())
        packed.append(UVd[2]._psgdk_lra.info()["packed_rows"])
    torch.cuda.synchronize()
    return [x.float().cpu() for x in UVd], hs, [float(x) for x in Luvd], packed


@pytest.mark.parametrize("r", [2, 4, 6, 8, 10, 12, 14, 16])
@pytest.mark.parametrize("N", [512, 5 * 512, 7 * 512 + 301, 300_001])
def test_packed_row_kernels_agree_with_the_one_row_kernels(N, r, monkeypatch):
    """Round 6: kernels_lra_pk.hiph (bf16 factors of even rank <= 16: one thread per row PAIR, rows packed in LDS) against
    kernels_lra.hiph on the same bf16 inputs -- every even rank the packed kernels are instantiated for, N = one block, whole blocks only,
    blocks + a ragged tail (the tail runs on the one-row kernels, partial sums meet in the same scratch slots), and a size where every
    workgroup loops.  PSGDK_LRA_PK=0 is the A/B switch.  The arithmetic of a row is the same in both; the reductions over rows sum in a
    different order, so the outputs differ by roundings of bf16 stores: both must sit equally close to the fp64 oracle on the same inputs
    (psgd.py:994-1072), and close to each other."""
    U0, V0, d0, gs, vs = _pk_inputs(N, r, seed=N % 97)
    monkeypatch.delenv("PSGDK_LRA_PK", raising=False)
    a, ha, la, pa = _pk_run(U0, V0, d0, gs, vs)
    assert pa == [N // 512 * 512] * 6, pa           # the packed kernels did run, over every whole block
    monkeypatch.setenv("PSGDK_LRA_PK", "0")
    b, hb, lb, pb = _pk_run(U0, V0, d0, gs, vs)
    assert pb == [0] * 6, pb
    U64 = [U0.double(), V0.double(), d0.double()]
    L64 = [torch.zeros([], dtype=torch.float64) for _ in range(3)]
    for t, (g, v, c) in enumerate(zip(gs, vs, (0.2, 0.8, 0.2))):
        orc.update_precond_lra_whiten(U64, L64, g.double(), v.double(), c, lr=0.2, betaL=0.9, damping=1e-9)
        h64 = orc.precond_grad_lra(U64, g.double())
        e_pk, e_one = relerr(ha[t], h64), relerr(hb[t], h64)
        assert torch.isfinite(ha[t]).all()
        assert e_pk <= 1.25 * e_one + 2e-3, (N, r, t, "h", e_pk, e_one)
        assert relerr(ha[t], hb[t]) <= 8e-3, (N, r, t, relerr(ha[t], hb[t]))
    for nm, x, y, truth in zip(("U", "V", "d"), a, b, U64):
        e_pk, e_one = relerr(x, truth), relerr(y, truth)
        assert e_pk <= 1.25 * e_one + 2e-3, (N, r, nm, e_pk, e_one)
        assert relerr(x, y) <= 8e-3, (N, r, nm, relerr(x, y))
    for k in range(3):
        assert abs(la[k] - lb[k]) <= 2e-2 * abs(lb[k]), (la, lb)
        assert abs(la[k] - float(L64[k])) <= 1.25 * abs(lb[k] - float(L64[k])) + 1e-2 * abs(float(L64[k])), (k, la, lb, [float(x) for x in L64])


def test_packed_row_kernels_eligibility():
    """Who takes the packed kernels (include/psgdk.h PSGDK_LRA_INFO_PACKED_ROWS): bf16 AND even rank in 2..16 AND N >= 512 AND N-vectors on
    4-byte boundaries.  Everything else runs the one-row kernels with unchanged results: a gradient 2 bytes into its allocation gives what
    the aligned copy gives under PSGDK_LRA_PK=0 (the same kernels on the same values; the reductions over rows end in fp32 atomics, whose
    order is the scheduler's: equal to fp32 rounding of the r-vectors, i.e. to a rare bf16 flip of a stored element)."""
    from psgd_torch_amd import lra
    for N, r, dt, want in ((4096, 10, torch.bfloat16, 4096), (4096, 9, torch.bfloat16, 0), (4096, 18, torch.bfloat16, 0),
                           (511, 10, torch.bfloat16, 0), (4096, 10, torch.float32, 0), (4096 + 77, 4, torch.bfloat16, 4096)):
        gen = torch.Generator().manual_seed(N + r)
        UVd = [(0.01 * torch.randn(N, r, generator=gen)).to(dt).to(DEV), (0.01 * torch.randn(N, r, generator=gen)).to(dt).to(DEV),
               torch.ones(N, 1, dtype=dt, device=DEV)]
        Luvd = [torch.zeros([], dtype=torch.float32, device=DEV) for _ in range(3)]
        g = torch.randn(N, 1, generator=gen).to(dt).to(DEV)
        lra.update_precond_lra_whiten(UVd, Luvd, g, lr=0.1, betaL=0.9, damping=1e-9, v_noise=torch.randn(N, 1, generator=gen).to(dt).to(DEV), coin=0.1)
        assert UVd[2]._psgdk_lra.info()["packed_rows"] == want, (N, r, dt)
        h = lra.precond_grad_lra(UVd, g)
        assert UVd[2]._psgdk_lra.info()["packed_rows"] == want, (N, r, dt)
        assert torch.isfinite(h.float()).all()
    U0, V0, d0, gs, vs = _pk_inputs(7 * 512 + 301, 10, seed=5)
    a, ha, la, pa = _pk_run(U0, V0, d0, gs, vs, offset_by_one_element=True)
    assert pa == [0] * 6, pa
    import os
    os.environ["PSGDK_LRA_PK"] = "0"
    try:
        b, hb, lb, pb = _pk_run(U0, V0, d0, gs, vs)
    finally:
        del os.environ["PSGDK_LRA_PK"]
    for x, y in zip(a + ha, b + hb):
        assert relerr(x, y) <= 1e-3, relerr(x, y)
        assert float((x != y).float().mean()) <= 0.02
    for x, y in zip(la, lb):
        assert abs(x - y) <= 1e-4 * abs(y), (la, lb)
