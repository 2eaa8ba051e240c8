"""
Full-size parity (run with -m gpu on an MI355X): the shapes bench.py TIMES are here CHECKED -- HIP through the C ABI against
the fp64 oracle with replayed noise, at the true sizes of BASELINE.json's configurations:

  * GPT-2-small (misc/gpt2.py GPTConfig defaults), bf16 preconditioner: (2304,768), (768,768), (768,3072), (50304,768) --
    768-wide dense factors on the 256x256 tiling, the split-K mode Gram at K = 50304, and BOTH routes of the spectral-norm
    bound (cooperative launch with S = 3 workgroups per factor / the multi-launch route), each against the oracle;
  * GPT-2-medium: (4096,1024), (1024,1024) -- 1024-wide factors, multi-launch route;
  * one transformer block of GPT-2-small + wpe through the batched KWNS4 engine (12 + 1 tensors in one plan);
  * config 1: (784,10) + (10,), max_skew = inf, fp32 (784-wide fp32 dense factor: the multi-launch route at that width);
  * ViT-B/16-scale LRA: N = 2*10^7 (and the true N = 86,543,080 when the host has the memory), r = 10, fp32.

Acceptance (relative Frobenius): fp32 <= 3e-5 per step vs the fp64 oracle (the fp32 reference's own error is ~1e-6);
bf16: error vs the fp64 oracle trajectory <= 1.5 x the bf16 ORACLE's own error vs that trajectory + 1 bf16 ulp (2 for L).
Reference: psgd.py:394-419, :46-93, :994-1072; wrapped_as_torch_optimizer_for_ddp.py:98-176.
"""
import os

import pytest
import torch

from helpers import P_of, relerr
from oracle import psgd_oracle as orc

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
ULP = 7.8125e-3       # bf16


def _amd():
    import psgd_torch_amd
    return psgd_torch_amd


def _structured(shape, T, seed, scale=0.3):
    """G_t = H_1 X H_2 with fixed SPD mixers (cond ~ 10^2), so that the factors actually move."""
    g = torch.Generator().manual_seed(seed)
    mixers = []
    for s in shape:
        k = min(s, 256)
        W = torch.randn(s, k, generator=g) / (k ** 0.5)
        mixers.append((0.5, W))
    out = []
    for _ in range(T):
        X = torch.randn(*shape, generator=g)
        for i, (a, W) in enumerate(mixers):          # X <- (a I + W W^T) x_i X, applied without forming s x s matrices
            Xm = torch.movedim(X, i, 0)
            flat = Xm.reshape(Xm.shape[0], -1)
            flat = a * flat + W @ (W.t() @ flat)
            X = torch.movedim(flat.reshape(Xm.shape), 0, i)
        out.append(scale * X)
    return out


def _dev_noise(nz):
    return ([nz.g_noise.to(DEV)], {(0, i): x.to(DEV) for i, x in enumerate(nz.spd) if x is not None},
            {(0, i): x.to(DEV) for i, x in enumerate(nz.skh) if x is not None})


def _seam_case(shape, dt, fused, steps=2, max_skew=1.0, lr=0.5, seed=0, monkeypatch=None):
    amd = _amd()
    if monkeypatch is not None:
        monkeypatch.setenv("PSGDK_NLB_FUSED", "1" if fused else "0")          # read when the plan is created
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 64)))
    Gs = _structured(shape, steps, 1000 + seed)
    QL, exprs = amd.init_kron(torch.zeros(shape, device=DEV, dtype=dt), Scale=1.0, max_skew=max_skew)
    if monkeypatch is not None:
        monkeypatch.delenv("PSGDK_NLB_FUSED")
    QL64, kinds = orc.init_kron(torch.zeros(shape, dtype=torch.float64), Scale=1.0, max_skew=max_skew)
    QLlo, _ = orc.init_kron(torch.zeros(shape, dtype=dt), Scale=1.0, max_skew=max_skew)
    gen = torch.Generator().manual_seed(77 + seed)
    worst = {}
    for t in range(steps):
        Gd = Gs[t].to(dt)
        nz = orc.KronNoise.draw(Gd, kinds, gen)
        amd.update_precond_kron_whiten_q0p5eq1p5(QL, exprs, Gd.to(DEV), lr=lr, betaL=0.9, damping=1e-9, noise=_dev_noise(nz),
                                                 balance=False)
        h = amd.precond_grad_kron(QL, exprs, Gd.to(DEV))
        n64 = orc.KronNoise(nz.g_noise.double(), [x.double() if x is not None else None for x in nz.spd],
                            [x.double() if x is not None else None for x in nz.skh], 0.5)
        orc.update_precond_kron_whiten_q0p5eq1p5(QL64, Gd.double(), n64, lr=lr, betaL=0.9, damping=1e-9)
        h64 = orc.precond_grad_kron(QL64[0], Gd.double())
        nlo = orc.KronNoise(nz.g_noise, nz.spd, nz.skh, 0.5)
        orc.update_precond_kron_whiten_q0p5eq1p5(QLlo, Gd, nlo, lr=lr, betaL=0.9, damping=1e-9)
        hlo = orc.precond_grad_kron(QLlo[0], Gd)
        checks = [("h", h, hlo, h64)]
        for i in range(len(QL[0])):
            checks.append((f"P{i}", P_of([QL[0][i]])[0], P_of([QLlo[0][i]])[0], P_of([QL64[0][i]])[0]))
            checks.append((f"L{i}", QL[1][i], QLlo[1][i], QL64[1][i]))
        for what, got, low, truth in checks:
            assert bool(torch.isfinite(torch.as_tensor(got).float()).all()), (shape, t, what, "non-finite")
            e_hip, e_ref = relerr(got, truth), relerr(low, truth)
            worst[what] = max(worst.get(what, 0.0), e_hip)
            if dt == torch.float32:
                assert e_hip <= 3e-5 * (t + 1), (shape, "fp32", t, what, e_hip, e_ref)
            else:
                floor = 2 * ULP if what.startswith("L") else ULP
                assert e_hip <= 1.5 * e_ref + floor, (shape, "bf16", t, what, e_hip, e_ref)
    # the factors must have moved (otherwise the comparison says nothing about the update)
    for i, q in enumerate(QL64[0]):
        ref = torch.eye(q.shape[0], dtype=torch.float64) if q.dim() == 2 else torch.ones_like(q)
        assert relerr(q, ref) > 1e-2, (shape, i, "factor did not move")
    eng = exprs[0]
    return eng.info(), worst


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("shape", [(2304, 768), (768, 768), (768, 3072)])
def test_gpt2_small_shapes_bf16_both_norm_bound_routes(shape, fused, monkeypatch):
    info, _ = _seam_case(shape, torch.bfloat16, fused, monkeypatch=monkeypatch)
    assert info["nlb_coop"] == fused, info        # the 768-wide bf16 factor is eligible for the cooperative launch
    assert info["max_dense_dim"] == 768


GEOM_FNS = {"EQ": "update_precond_kron_whiten_eq", "QEQ": "update_precond_kron_whiten_qeq", "QUAD": "update_precond_kron_whiten_quad",
            "QEP": "update_precond_kron_whiten_qep", "QUAD4P": "update_precond_kron_whiten_quad4p",
            "PRO4P": "update_precond_kron_whiten_pro4p"}


@pytest.mark.parametrize("shape,geom", [((2304, 768), "EQ"), ((768, 768), "EQ"), ((3072, 768), "EQ"), ((2304, 768), "QEQ"),
                                        ((768, 768), "QUAD"), ((2304, 768), "QEP"), ((768, 768), "QUAD4P"), ((1536, 768), "PRO4P")])
def test_gpt2_small_shapes_other_geometries_bf16(shape, geom):
    """The other fitting geometries at GPT-2-small's sizes, bf16, vs the fp64 oracle with replayed noise: for EQ that is the bf16
    triangular solve with the panel in LDS at d = 768 (both solves for (768,768): row and column factor dense), the K-band
    skipping of the triangular GEMM operands, and Q' = Q - mu triu(.) Q staying exactly upper triangular (psgd.py:278-336);
    QEQ / QUAD / QEP / QUAD4P share Pg, the Grams and the norm bound with the default geometry (psgd.py:339-391, 455-513)."""
    amd = _amd()
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 64)))
    dt = torch.float32 if geom == "PRO4P" else torch.bfloat16     # (PRO4P's data-dependent number of rotations: fp32, as in the fuzz test)
    p4 = geom in ("QUAD4P", "PRO4P")
    upd_amd, upd_orc = getattr(amd, GEOM_FNS[geom]), getattr(orc, GEOM_FNS[geom])
    Gs = _structured(shape, 2, 2000 + len(geom))
    QL, exprs = amd.init_kron(torch.zeros(shape, device=DEV, dtype=dt), Scale=1.0, max_skew=1.0, dQ=geom)
    QL64, kinds = orc.init_kron(torch.zeros(shape, dtype=torch.float64), Scale=1.0, max_skew=1.0)
    QLlo, _ = orc.init_kron(torch.zeros(shape, dtype=dt), Scale=1.0, max_skew=1.0)
    gen = torch.Generator().manual_seed(91)
    for t in range(2):
        Gd = Gs[t].to(dt)
        nz = orc.KronNoise.draw(Gd, kinds, gen)
        nz.balance_u = 1.0
        dn = _dev_noise(nz)
        pro = pro64 = None
        if geom == "PRO4P":       # ten draws per dense factor for the successive procrustes_step3 calls (psgd.py:422-452), stacked for the ABI
            pro = [None if x is None else [torch.randn(x.shape, generator=gen).to(dt) for _ in range(10)] for x in nz.skh]
            pro64 = [None if p_ is None else [y.double() for y in p_] for p_ in pro]
            dn = (dn[0], dn[1], {(0, i): torch.cat(p_, dim=0).to(DEV) for i, p_ in enumerate(pro) if p_ is not None})
        kwargs = dict(lr=0.3, betaL=0.9, damping=1e-9, noise=dn)
        if geom != "QEP":
            kwargs["balance"] = False
        upd_amd(QL, exprs, Gd.to(DEV), **kwargs)
        h = amd.precond_grad_kron(QL, exprs, Gd.to(DEV))
        n64 = orc.KronNoise(nz.g_noise.double(), [x.double() if x is not None else None for x in nz.spd],
                            [x.double() if x is not None else None for x in nz.skh], 1.0)
        if geom == "PRO4P":
            upd_orc(QL64, Gd.double(), n64, pro64, lr=0.3, betaL=0.9, damping=1e-9)
            upd_orc(QLlo, Gd, orc.KronNoise(nz.g_noise, nz.spd, nz.skh, 1.0), pro, lr=0.3, betaL=0.9, damping=1e-9)
        else:
            upd_orc(QL64, Gd.double(), n64, lr=0.3, betaL=0.9, damping=1e-9)
            upd_orc(QLlo, Gd, orc.KronNoise(nz.g_noise, nz.spd, nz.skh, 1.0), lr=0.3, betaL=0.9, damping=1e-9)
        ap = orc.precond_grad_kron_4p if p4 else orc.precond_grad_kron
        checks = [("h", h, ap(QLlo[0], Gd), ap(QL64[0], Gd.double()))]
        for i in range(len(QL[0])):
            checks.append((f"Q{i}", QL[0][i], QLlo[0][i], QL64[0][i]))       # no gauge freedom in these geometries: Q itself
            checks.append((f"L{i}", QL[1][i], QLlo[1][i], QL64[1][i]))
            if geom == "EQ" and QL[0][i].dim() == 2:
                assert float(torch.tril(QL[0][i].float(), -1).abs().max()) == 0.0, (shape, i, "Q left the upper triangle")
        for what, got, low, truth in checks:
            assert bool(torch.isfinite(torch.as_tensor(got).float()).all()), (shape, geom, t, what, "non-finite")
            e_hip, e_ref = relerr(got, truth), relerr(low, truth)
            floor = 2 * ULP if what.startswith("L") else ULP
            if dt == torch.float32:          # (rotations of a fitted P amplify fp32 rounding: psgd.py:425-426; same bound as the fuzz test)
                assert e_hip <= 2e-3, (shape, geom, t, what, e_hip, e_ref)
            else:
                assert e_hip <= 1.5 * e_ref + floor, (shape, geom, t, what, e_hip, e_ref)
    for i, q in enumerate(QL64[0]):
        ref = torch.eye(q.shape[0], dtype=torch.float64) if q.dim() == 2 else torch.ones_like(q)
        assert relerr(q, ref) > 1e-2, (shape, geom, i, "factor did not move")


@pytest.mark.parametrize("fused", [1, 0])
def test_gpt2_small_wte_bf16_splitk_gram(fused, monkeypatch):
    """wte (50304, 768): the mode Gram contracts over K = 50304 (split-K slabs + splitk_reduce_sym), 197 row tiles."""
    info, _ = _seam_case((50304, 768), torch.bfloat16, fused, monkeypatch=monkeypatch, seed=3)
    assert info["nlb_coop"] == fused, info


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("shape", [(4096, 1024), (1024, 1024), (1024, 4096)])
def test_gpt2_medium_shapes_bf16(shape, fused, monkeypatch):
    """GPT-2-medium's 1024-wide factors on both norm-bound routes: the multi-launch one (what the whole 123-factor plan runs) and, since round 6,
    the cooperative launch with 32 K steps of registers and members of 128 columns (plans of up to 31 such factors: a rank's share)."""
    info, _ = _seam_case(shape, torch.bfloat16, fused, monkeypatch=monkeypatch, seed=5)
    assert info["nlb_coop"] == fused and info["max_dense_dim"] == 1024, info
    if fused:
        assert info["nlb_member_cols"] == 128, info


def test_gpt2_small_shape_fp32():
    """The same path in fp32 (f32 MFMA, 768 > 384: multi-launch norm bound) at full width: tight bound vs the fp64 oracle."""
    info, worst = _seam_case((2304, 768), torch.float32, None, seed=7)
    assert info["nlb_coop"] == 0, info


def _kwns4_vs_oracle(shapes, kw, steps, tol_fp32=None, seed=0, grad_scale=0.3):
    amd = _amd()
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 64)))
    gen = torch.Generator().manual_seed(seed)
    p_cpu = [0.02 * torch.randn(s, generator=gen) for s in shapes]
    params = [torch.nn.Parameter(p.clone().to(DEV)) for p in p_cpu]
    opt = amd.KWNS4(params, **kw)
    cur = {}

    def noise_for(G, kinds):
        n = orc.KronNoise.draw(G, kinds, gen)
        cur.setdefault("list", []).append(n)
        return n

    def mk_oracle(pdt):
        k2 = dict(kw)
        k2["preconditioner_dtype"] = pdt
        return k2

    ora = orc.KWNS4Oracle([p.clone() for p in p_cpu], uniform=lambda: 0.0, noise_for=noise_for, **kw)
    # the fp64 "truth" run replays the same draws, upcast
    it = {}

    def noise_for64(G, kinds):
        n = it["list"].pop(0)
        return orc.KronNoise(n.g_noise.double(), [x.double() if x is not None else None for x in n.spd],
                             [x.double() if x is not None else None for x in n.skh], n.balance_u)

    kw64 = dict(kw)
    ora64 = orc.KWNS4Oracle([p.double() for p in p_cpu], uniform=lambda: 0.0, noise_for=noise_for64, **kw64)
    ora64.g["preconditioner_dtype"] = torch.float64
    for step in range(steps):
        grads = [g for g in (_structured(s, 1, 31 * step + 7 * i + seed, scale=grad_scale)[0] if len(s) == 2 else
                             grad_scale * torch.randn(s, generator=gen) for i, s in enumerate(shapes))]
        cur["list"] = []
        ora.step([g.clone() for g in grads])
        per = cur["list"]
        it["list"] = list(per)
        ora64.step([g.double() for g in grads])

        def replay(b, plist, per=per):
            g = [per[i].g_noise.to(DEV) for i in b.owned]
            spd = {(k, j): x.to(DEV) for k, i in enumerate(b.owned) for j, x in enumerate(per[i].spd) if x is not None}
            skh = {(k, j): x.to(DEV) for k, i in enumerate(b.owned) for j, x in enumerate(per[i].skh) if x is not None}
            return dict(noise=(g, spd, skh), balance_mask=[per[i].balance_u < 0.01 for i in b.owned])

        opt._uniform = lambda: 0.0
        opt._replay = replay
        for p, g in zip(params, grads):
            p.grad = g.to(DEV)
        opt.step()
    torch.cuda.synchronize()
    out = []
    for k, (p, q, q64, p0) in enumerate(zip(params, ora.params, ora64.params, p_cpu)):
        # compare the UPDATE (p - p0): the parameters themselves are dominated by their initial values
        d_hip = p.detach().cpu().double() - p0.double()
        d_lo = q.double() - p0.double()
        d_64 = q64 - p0.double()
        out.append((shapes[k], relerr(d_hip, d_64), relerr(d_lo, d_64)))
        assert bool(torch.isfinite(p).all())
    return opt, out


GPT2_BLOCK = [(1024, 768), (768,), (768,), (2304, 768), (2304,), (768, 768), (768,), (768,), (768,), (3072, 768), (3072,),
              (768, 3072), (768,)]


def test_gpt2_small_block_through_batched_kwns4_bf16():
    """wpe + one transformer block (misc/gpt2.py:116-118,187-189,215-227) in ONE plan, KWNS4 defaults (bf16 preconditioner,
    momentum 0.9, whiten momentum), 3 steps: the accumulated parameter update vs the fp64 oracle run of the same steps."""
    kw = dict(preconditioner_dtype=torch.bfloat16, lr_params=1e-3, weight_decay=0.0)
    opt, res = _kwns4_vs_oracle(GPT2_BLOCK, kw, steps=3, seed=11)
    for shape, e_hip, e_ref in res:
        assert e_hip <= 1.5 * e_ref + ULP, (shape, e_hip, e_ref)
    eng = next(iter(opt._buckets.values())).engine
    assert eng.info()["nlb_coop"] == 1 and eng.info()["nlb_fallbacks"] == 0, eng.info()


def test_config1_logistic_regression_shapes_fp32():
    """BASELINE config 1: (784,10) + (10,), Kron with max_skew = inf (both dims of the weight dense: 784^2 and 10^2), fp32,
    5 KWNS4 steps vs the fp64 oracle.  The 784-wide fp32 factor takes the multi-launch norm-bound route."""
    kw = dict(preconditioner_dtype=torch.float32, preconditioner_max_skew=float("inf"), lr_params=1e-2, weight_decay=0.0)
    opt, res = _kwns4_vs_oracle([(784, 10), (10,)], kw, steps=5, seed=21)
    for shape, e_hip, e_ref in res:
        assert e_hip <= 2e-4, (shape, e_hip, e_ref)
    eng = next(iter(opt._buckets.values())).engine
    assert eng.info()["nlb_coop"] == 0 and eng.info()["max_dense_dim"] == 832, eng.info()


def _lra_case(N, r, steps=2, seed=0):
    amd = _amd()
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 64)))
    g = torch.Generator().manual_seed(90 + seed)
    dt = torch.float32
    U0 = torch.randn(N, r, generator=g)
    U0 *= 0.1 ** 0.5 / float(torch.linalg.vector_norm(U0))
    V0 = torch.randn(N, r, generator=g)
    V0 *= 0.1 ** 0.5 / float(torch.linalg.vector_norm(V0))
    d0 = 0.5 + torch.rand(N, 1, generator=g)
    hscale = 0.5 + 2 * torch.rand(N, 1, generator=g)
    UVd = [U0.clone().to(DEV), V0.clone().to(DEV), d0.clone().to(DEV)]
    Luvd = [torch.zeros([], dtype=torch.float32, device=DEV) for _ in range(3)]
    UVd64 = [U0.double(), V0.double(), d0.double()]
    Luvd64 = [torch.zeros([], dtype=torch.float64) for _ in range(3)]
    for t in range(steps):
        gt = hscale * torch.randn(N, 1, generator=g)
        vn = torch.randn(N, 1, generator=g)
        coin = 0.25 if t % 2 == 0 else 0.75
        amd.update_precond_lra_whiten(UVd, Luvd, gt.to(DEV), lr=0.1, betaL=0.9, damping=1e-9, v_noise=vn.to(DEV), coin=coin)
        h = amd.precond_grad_lra(UVd, gt.to(DEV))
        orc.update_precond_lra_whiten(UVd64, Luvd64, gt.double(), vn.double(), coin, lr=0.1, betaL=0.9, damping=1e-9)
        h64 = orc.precond_grad_lra(UVd64, gt.double())
        for what, got, truth in (("h", h, h64), ("U", UVd[0], UVd64[0]), ("V", UVd[1], UVd64[1]), ("d", UVd[2], UVd64[2])):
            e = relerr(got, truth)
            assert e <= 5e-5 * (t + 1), (N, r, t, what, e)
        for k in range(3):
            assert relerr(Luvd[k], Luvd64[k]) <= 5e-5 * (t + 1), (N, t, "L", k)
    assert relerr(UVd64[0], U0.double()) > 1e-3 or relerr(UVd64[1], V0.double()) > 1e-3


def test_lra_vit_b_scale_n2e7_r10():
    """ViT-B/16-scale LRA (one concatenated vector), N = 2*10^7, r = 10, fp32 vs the fp64 oracle
    (psgd.py:994-1072)."""
    _lra_case(20_000_003, 10)          # ragged N (not a multiple of any tile)


@pytest.mark.parametrize("r", [24, 32, 48, 64])
def test_lra_wide_ranks_n3e6(r):
    """The two wider rank classes (two / four threads per row: 128- / 64-row blocks, tiled Grams, wave-parallel LU) at a size
    where every workgroup loops several times, ragged N; fp32 vs the fp64 oracle (psgd.py:994-1072)."""
    _lra_case(3_000_017, r, steps=2, seed=r)


def test_lra_vit_b_true_n_r10():
    """The true ViT-B/16 N = 86,543,080 (misc/vit.py ViT(224,16,1000,768,12,12,3072)); needs ~25 GB of host memory for
    the fp64 oracle."""
    try:
        import psutil
        if psutil.virtual_memory().available < 40 * 2 ** 30:
            pytest.skip("not enough host memory for the fp64 oracle at N = 86.5 M")
    except ImportError:
        pass
    _lra_case(86_543_080, 10, steps=1, seed=1)


def _lra_case_bf16(N, r, steps=2, seed=0, with_bf16_oracle=True):
    """bf16 factors (the reference's LRAWhiten keeps U, V, d in the parameter dtype: psgd.py:1118-1128) on the HIP path vs the fp64 oracle on
    the SAME bf16-rounded inputs; the yardstick is the oracle run in bf16 (torch CPU bf16 arithmetic = what the reference would compute):
    HIP error <= 1.5 x its error + one bf16 ulp.  Without the bf16 oracle (true N: host memory): absolute bounds a few ulp wide."""
    amd = _amd()
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 64)))
    g = torch.Generator().manual_seed(190 + seed)
    bf = torch.bfloat16
    U0 = torch.randn(N, r, generator=g)
    U0 *= 0.1 ** 0.5 / float(torch.linalg.vector_norm(U0))
    U0 = U0.to(bf)
    V0 = torch.randn(N, r, generator=g)
    V0 *= 0.1 ** 0.5 / float(torch.linalg.vector_norm(V0))
    V0 = V0.to(bf)
    d0 = (0.5 + torch.rand(N, 1, generator=g)).to(bf)
    hscale = 0.5 + 2 * torch.rand(N, 1, generator=g)
    UVd = [U0.clone().to(DEV), V0.clone().to(DEV), d0.clone().to(DEV)]
    Luvd = [torch.zeros([], dtype=torch.float32, device=DEV) for _ in range(3)]
    UVd64 = [U0.double(), V0.double(), d0.double()]
    Luvd64 = [torch.zeros([], dtype=torch.float64) for _ in range(3)]
    UVdb = [U0.clone(), V0.clone(), d0.clone()] if with_bf16_oracle else None
    Luvdb = [torch.zeros([], dtype=bf) for _ in range(3)]
    for t in range(steps):
        gt = (hscale * torch.randn(N, 1, generator=g)).to(bf)
        vn = torch.randn(N, 1, generator=g).to(bf)
        coin = 0.25 if t % 2 == 0 else 0.75
        amd.update_precond_lra_whiten(UVd, Luvd, gt.to(DEV), lr=0.1, betaL=0.9, damping=1e-9, v_noise=vn.to(DEV), coin=coin)
        assert UVd[2]._psgdk_lra.info()["packed_rows"] == (N // 512 * 512 if (r % 2 == 0 and 2 <= r <= 16) else 0)
        h = amd.precond_grad_lra(UVd, gt.to(DEV))
        orc.update_precond_lra_whiten(UVd64, Luvd64, gt.double(), vn.double(), coin, lr=0.1, betaL=0.9, damping=1e-9)
        h64 = orc.precond_grad_lra(UVd64, gt.double())
        if with_bf16_oracle:
            orc.update_precond_lra_whiten(UVdb, Luvdb, gt, vn, coin, lr=0.1, betaL=0.9, damping=1e-9)
            hb = orc.precond_grad_lra(UVdb, gt)
            refs = (hb, UVdb[0], UVdb[1], UVdb[2])
        else:
            refs = (None,) * 4
        for what, got, truth, ref in zip(("h", "U", "V", "d"), (h, UVd[0], UVd[1], UVd[2]), (h64, UVd64[0], UVd64[1], UVd64[2]), refs):
            assert torch.isfinite(got.float()).all()
            e = relerr(got, truth)
            if ref is not None:
                e_ref = relerr(ref, truth)
                assert e <= 1.5 * e_ref + ULP, (N, r, t, what, e, e_ref)
            else:
                assert e <= 2.5 * ULP * (t + 1), (N, r, t, what, e)
        for k in range(3):
            e = relerr(Luvd[k], Luvd64[k])
            if with_bf16_oracle:
                assert e <= 1.5 * relerr(Luvdb[k], Luvd64[k]) + 2 * ULP, (N, t, "L", k, e)
            else:
                assert e <= 5 * ULP, (N, t, "L", k, e)
    assert relerr(UVd64[0], U0.double()) > 1e-3 or relerr(UVd64[1], V0.double()) > 1e-3


@pytest.mark.parametrize("r", [10, 16, 4])
def test_lra_bf16_vit_b_scale_n2e7(r):
    """ViT-B/16-scale LRA in bf16 (SURVEY section 8d quotes config 4 in both dtypes), N = 2*10^7 ragged, on the packed two-rows-per-thread
    kernels (kernels_lra_pk.hiph; the 387-row tail on the one-row kernels) vs the fp64 oracle, yardstick: the oracle in bf16
    (psgd.py:994-1072)."""
    _lra_case_bf16(20_000_387, r, steps=2, seed=r)


def test_lra_bf16_odd_rank_n3e6():
    """an odd rank keeps the one-row kernels (bf16)"""
    _lra_case_bf16(3_000_017, 7, steps=2, seed=7)


def test_lra_bf16_vit_b_true_n_r10():
    """The true ViT-B/16 N = 86,543,080 in bf16, r = 10, one update + apply vs the fp64 oracle (absolute bounds: no bf16 oracle at this
    size, host memory)."""
    try:
        import psutil
        if psutil.virtual_memory().available < 40 * 2 ** 30:
            pytest.skip("not enough host memory for the fp64 oracle at N = 86.5 M")
    except ImportError:
        pass
    _lra_case_bf16(86_543_080, 10, steps=1, seed=1, with_bf16_oracle=False)
