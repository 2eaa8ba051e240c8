"""Randomised differential test of the HIP path against the CPU oracle (fp32, explicit noise): random shapes (odd sizes,
singleton dims, 0..4 dims), dense/diagonal decisions (max_skew, max_size), all seven geometries, a few steps each.
Catches layout / edge-tile / padding mistakes the fixed golden shapes may miss; the oracle is pinned to the reference by
tests/test_oracle_golden.py."""
import random

import pytest
import torch

from helpers import P_of, relerr
from oracle import psgd_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

GEOMS = ["Q0.5EQ1.5", "EQ", "QEQ", "QUAD", "QEP", "QUAD4P", "PRO4P"]


def _case(seed):
    rnd = random.Random(seed)
    nd = rnd.choice([0, 1, 1, 2, 2, 2, 2, 3, 4])
    if nd <= 2:
        shape = tuple(rnd.choice([1, 2, 3, 7, 16, 33, 63, 64, 65, 100, 129, 200, 257]) for _ in range(nd))
    else:
        shape = tuple(rnd.choice([1, 2, 3, 5, 8, 12]) for _ in range(nd))
    max_skew = rnd.choice([0.0, 0.5, 1.0, 2.0, float("inf")])
    max_size = rnd.choice([float("inf"), float("inf"), 20.0, 100.0])
    geom = rnd.choice(GEOMS)
    return shape, max_skew, max_size, geom


import os  # noqa: E402

N_CASES = int(os.environ.get("PSGDK_FUZZ_CASES", "40"))        # raise for a longer hunt (e.g. PSGDK_FUZZ_CASES=1000)


@pytest.mark.parametrize("seed", list(range(N_CASES)))
def test_random_case_matches_oracle(seed):
    _run_case(seed, *_case(seed))


@pytest.mark.parametrize("shape,geom", [((96, 5000), "Q0.5EQ1.5"), ((5000, 96), "Q0.5EQ1.5"), ((70, 9000), "QEQ"), ((4500, 128), "PRO4P")])
def test_long_contracted_dimension_split_k(shape, geom):
    """A mode Gram whose contracted extent exceeds 4096 is computed as split-K partial sums (fp32 slabs) and finished by
    splitk_reduce_sym_kernel -- the wte path of GPT-2 (K = 50304).  The long dimension is kept diagonal (max_size), so the
    dense factor is small and the oracle stays cheap; K = 5056 / 9024 / 4544 gives 2 / 3 / 2 slabs of 3072."""
    _run_case(900 + len(geom) + shape[0], shape, float("inf"), 4096.0, geom)


@pytest.mark.parametrize("shape,geom,max_skew", [((2,) * 9, "Q0.5EQ1.5", 1.0), ((2, 3) * 5, "QEQ", 1.0), ((2,) * 12, "Q0.5EQ1.5", 1.0),
                                                 ((3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2), "EQ", 1.0),
                                                 ((2,) * 16, "QUAD", float("inf")), ((2,) * 9, "PRO4P", 1.0)])
def test_tensors_with_9_to_16_dims(shape, geom, max_skew):
    """The reference allows up to 26 dims (one einsum letter each, psgd.py:197-198); the mode-product path takes them all.
    (26 dims of extent 2 would be 2^26 elements with 26 factors: same code, left to the plan-creation test on CPU.)"""
    _run_case(7000 + len(shape), shape, max_skew, float("inf"), geom)


@pytest.mark.parametrize("ndim,geom,steps", [(20, "Q0.5EQ1.5", 2), (20, "QUAD", 2), (26, "Q0.5EQ1.5", 1)])
def test_tensors_with_20_and_26_dims(ndim, geom, steps):
    """The reference's limit itself: 26 dims (one einsum letter each, psgd.py:197-198) of extent 2 = 2^26 elements with 26 dense 2 x 2
    factors (4 <= numel: all dense at max_skew = 1), and 20 dims; PSGDK_MAX_DIMS noise slots per tensor are all in use.  The
    26-dim case moves 67 M elements through 26 mode products and 26 mode Grams on both sides: there every mode Gram is a sum of 2^25
    products, where an fp32 reference carries its own rounding -- so that case is judged against the fp64 oracle on the same draws, by the
    criterion of the bf16 tests (error vs fp64 <= 1.5 x the fp32 oracle's own + the usual bound).  PSGDK_SKIP_26DIM=1 leaves it out."""
    if ndim == 26 and os.environ.get("PSGDK_SKIP_26DIM", "0") == "1":
        pytest.skip("2^26-element 26-dim tensor: PSGDK_SKIP_26DIM=1")
    # the CPU oracle's strided 26-dim copies thrash when torch spreads them over every core of a 128-core host (minutes instead of
    # the ~20 s they take on 8 threads); the GPU side of this case takes 0.1 s (tools/diag_26dim.py)
    n = torch.get_num_threads()
    torch.set_num_threads(min(n, 16))
    try:
        _run_case(7100 + ndim, (2,) * ndim, 1.0, float("inf"), geom, steps=steps, truth64=(ndim == 26))
    finally:
        torch.set_num_threads(n)


def _run_case(seed, shape, max_skew, max_size, geom, steps=3, truth64=False):
    import psgd_torch_amd as amd
    sq = tuple(s for s in shape if s != 1)                     # the wrappers squeeze first (..._ddp.py:124)
    upd_amd = {"Q0.5EQ1.5": amd.update_precond_kron_whiten_q0p5eq1p5, "EQ": amd.update_precond_kron_whiten_eq,
               "QEQ": amd.update_precond_kron_whiten_qeq, "QUAD": amd.update_precond_kron_whiten_quad,
               "QEP": amd.update_precond_kron_whiten_qep, "QUAD4P": amd.update_precond_kron_whiten_quad4p,
               "PRO4P": amd.update_precond_kron_whiten_pro4p}[geom]
    upd_orc = {"Q0.5EQ1.5": orc.update_precond_kron_whiten_q0p5eq1p5, "EQ": orc.update_precond_kron_whiten_eq,
               "QEQ": orc.update_precond_kron_whiten_qeq, "QUAD": orc.update_precond_kron_whiten_quad,
               "QEP": orc.update_precond_kron_whiten_qep, "QUAD4P": orc.update_precond_kron_whiten_quad4p,
               "PRO4P": orc.update_precond_kron_whiten_pro4p}[geom]
    p4 = geom in ("QUAD4P", "PRO4P")
    tol = 2e-3 if geom == "PRO4P" else 2e-4        # fitting P with repeated rotations amplifies fp32 rounding (psgd.py:425-426)
    gen = torch.Generator().manual_seed(1000 + seed)
    kw = dict(Scale=0.7, max_size=max_size, max_skew=max_skew)
    QL, exprs = amd.init_kron(torch.zeros(sq, device=DEV), dQ=geom, **kw)
    QLo, kinds = orc.init_kron(torch.zeros(sq), **(dict(kw, Scale=0.7 ** 2) if p4 else kw))   # psgd.py:186-187
    QL64 = orc.init_kron(torch.zeros(sq, dtype=torch.float64), **(dict(kw, Scale=0.7 ** 2) if p4 else kw))[0] if truth64 else None
    assert [q.dim() == 2 for q in QL[0]] == [k == "dense" for k in kinds], (shape, kinds)
    for t in range(steps):
        G = 0.5 * torch.randn(sq, generator=gen)
        nz = orc.KronNoise.draw(G, kinds, gen)
        nz.balance_u = 0.0 if t == 1 else 1.0                  # exercise the balancing branch once
        skh = {(0, i): x.to(DEV) for i, x in enumerate(nz.skh) if x is not None}
        pro = None
        if geom == "PRO4P":       # ten draws per dense factor for the successive procrustes_step3 calls, stacked for the ABI
            pro = [None if x is None else [torch.randn(x.shape, generator=gen) for _ in range(10)] for x in nz.skh]
            skh = {(0, i): torch.cat(p, dim=0).to(DEV) for i, p in enumerate(pro) if p is not None}
        dev_noise = ([nz.g_noise.to(DEV)], {(0, i): x.to(DEV) for i, x in enumerate(nz.spd) if x is not None}, skh)
        kwargs = dict(lr=0.2, betaL=0.9, damping=1e-6, noise=dev_noise)
        if geom != "QEP":
            kwargs["balance"] = nz.balance_u < 0.01
        upd_amd(QL, exprs, G.to(DEV), **kwargs)
        if geom == "PRO4P":
            upd_orc(QLo, G, nz, pro, lr=0.2, betaL=0.9, damping=1e-6)
        else:
            upd_orc(QLo, G, nz, lr=0.2, betaL=0.9, damping=1e-6)
        h = amd.precond_grad_kron(QL, exprs, G.to(DEV))
        ho = orc.precond_grad_kron_4p(QLo[0], G) if p4 else orc.precond_grad_kron(QLo[0], G)
        tag = (seed, shape, max_skew, max_size, geom, t)
        if truth64:
            # the same draws through the fp64 oracle = truth; the engine may be as far from it as 1.5 x the fp32 oracle is (+ the bound)
            assert geom == "Q0.5EQ1.5"
            nz64 = orc.KronNoise(nz.g_noise.double(), [None if x is None else x.double() for x in nz.spd],
                                 [None if x is None else x.double() for x in nz.skh], nz.balance_u)
            upd_orc(QL64, G.double(), nz64, lr=0.2, betaL=0.9, damping=1e-6)
            h64 = orc.precond_grad_kron(QL64[0], G.double())
            e_hip, e_ref = relerr(h, h64), relerr(ho, h64)
            assert e_hip <= 1.5 * e_ref + tol, tag + ("h vs fp64", e_hip, e_ref)
            for i in range(len(QL[0])):
                e_hip, e_ref = relerr(P_of([QL[0][i]])[0], P_of([QL64[0][i]])[0]), relerr(P_of([QLo[0][i]])[0], P_of([QL64[0][i]])[0])
                assert e_hip <= 1.5 * e_ref + tol, tag + (i, "P vs fp64", e_hip, e_ref)
                e_hip, e_ref = relerr(QL[1][i], QL64[1][i]), relerr(QLo[1][i], QL64[1][i])
                assert e_hip <= 1.5 * e_ref + tol, tag + (i, "L vs fp64", e_hip, e_ref)
            continue
        # (errors into plain floats BEFORE the assert: on a failure pytest's rewritten assert prints the repr of every operand, and the
        #  repr of a 26-dim tensor walks 2^26 Python frames -- the "hang" of the 26-dim case in round 3 was a failing assert being printed)
        e_h = relerr(h, ho)
        assert e_h <= tol, tag + ("h", e_h)
        for i in range(len(QL[0])):
            if geom == "Q0.5EQ1.5":                            # Q is gauge dependent there (Procrustes on rounding noise)
                e_q = relerr(P_of([QL[0][i]])[0], P_of([QLo[0][i]])[0])
                assert e_q <= tol, tag + (i, "P", e_q)
            else:
                e_q = relerr(QL[0][i], QLo[0][i])
                assert e_q <= tol, tag + (i, "Q", e_q)
            e_l = relerr(QL[1][i], QLo[1][i])
            assert e_l <= tol, tag + (i, "L", e_l)


M_CASES = int(os.environ.get("PSGDK_FUZZ_KWNS4", "12"))


@pytest.mark.parametrize("seed", list(range(M_CASES)))
def test_random_kwns4_configuration_matches_oracle_loop(seed):
    """The batched optimizer (several tensors of random shapes in ONE engine, random hyper-parameters) against the oracle's
    restatement of the reference loop (wrapped_as_torch_optimizer_for_ddp.py:98-176), noise replayed, fp32."""
    import psgd_torch_amd as amd
    rnd = random.Random(5000 + seed)
    n = rnd.randint(2, 7)
    shapes = []
    for _ in range(n):
        nd = rnd.choice([0, 1, 2, 2, 2, 3])
        shapes.append(tuple(rnd.choice([1, 3, 8, 17, 40, 64, 70, 130] if nd <= 2 else [1, 2, 4, 6]) for _ in range(nd)))
    momentum = rnd.choice([0.0, 0.9, 0.5])
    kw = dict(preconditioner_dtype=torch.float32, lr_params=1e-2, lr_preconditioner=rnd.choice([0.1, 0.5]),
              momentum=momentum, whiten_grad=(rnd.random() < 0.5) or momentum == 0.0,
              weight_decay=rnd.choice([0.0, 0.05]), decoupled_weight_decay=rnd.random() < 0.5,
              update_preconditioner_first=rnd.random() < 0.5, preconditioner_max_skew=rnd.choice([1.0, 2.0, float("inf")]),
              preconditioner_max_size=rnd.choice([float("inf"), 50.0]), preconditioner_init_scale=rnd.choice([1.0, 0.3]),
              grad_clip_max_amps=rnd.choice([(2.0, 10.0), (1.0, 1.5)]))
    gen = torch.Generator().manual_seed(6000 + seed)
    p_cpu = [0.3 * torch.randn(s, generator=gen) for s in shapes]
    params = [torch.nn.Parameter(p.clone().to(DEV)) for p in p_cpu]
    opt = amd.KWNS4(params, **kw)
    cur = {}

    def noise_for(G, kinds):
        nz = orc.KronNoise.draw(G, kinds, gen)
        cur.setdefault("list", []).append(nz)
        return nz
    ref_p = [p.clone() for p in p_cpu]
    oracle = orc.KWNS4Oracle(ref_p, uniform=lambda: 0.0, noise_for=noise_for, **kw)
    for step in range(3):
        grads = [0.4 * torch.randn(s, generator=gen) for s in shapes]
        cur["list"] = []
        oracle.step([g.clone() for g in grads])
        per = cur["list"]

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
    for i, (p, q) in enumerate(zip(params, ref_p)):
        assert relerr(p.detach(), q) <= 1e-4, (seed, shapes, kw, i, relerr(p.detach(), q))


L_CASES = int(os.environ.get("PSGDK_FUZZ_LRA", "24"))


@pytest.mark.parametrize("seed", list(range(L_CASES)))
def test_random_lra_case_matches_oracle(seed):
    """LRA update + apply with random N (ragged against the 256- / 128- / 64-row blocks of the three rank classes), rank 0..64 and,
    through the general path, up to 200; both update branches; fp32."""
    from psgd_torch_amd import lra
    rnd = random.Random(9000 + seed)
    N = rnd.choice([1, 2, 17, 63, 65, 127, 129, 255, 256, 257, 511, 1000, 2049, 5000])
    r = min(rnd.choice([0, 1, 2, 5, 10, 16, 17, 24, 32, 33, 47, 64, 65, 100, 129, 200]), max(N - 1, 0))
    gen = torch.Generator().manual_seed(9100 + seed)
    U = torch.randn(N, r, generator=gen); V = torch.randn(N, r, generator=gen)
    if r:
        U = U * (0.1 ** 0.5 / torch.linalg.vector_norm(U)); V = V * (0.1 ** 0.5 / torch.linalg.vector_norm(V))
    d = 0.5 + torch.rand(N, 1, generator=gen)
    UVd = [U.clone().to(DEV).contiguous(), V.clone().to(DEV).contiguous(), d.clone().to(DEV).contiguous()]
    Luvd = [torch.zeros([], device=DEV) for _ in range(3)]
    Uo = [U.clone(), V.clone(), d.clone()]
    Lo = [torch.zeros([]) for _ in range(3)]
    for t in range(3):
        g = (0.5 + 2 * torch.rand(N, 1, generator=gen)) * torch.randn(N, 1, generator=gen)
        vn = torch.randn(N, 1, generator=gen)
        coin = 0.25 if (t + seed) % 2 == 0 else 0.75
        lra.update_precond_lra_whiten(UVd, Luvd, g.to(DEV), lr=0.1, betaL=0.9, damping=1e-6, v_noise=vn.to(DEV), coin=coin)
        orc.update_precond_lra_whiten(Uo, Lo, g, vn, coin, lr=0.1, betaL=0.9, damping=1e-6)
        h = lra.precond_grad_lra(UVd, g.to(DEV))
        ho = orc.precond_grad_lra(Uo, g)
        tag = (seed, N, r, t)
        assert relerr(h, ho) <= 2e-4, tag + ("h", relerr(h, ho))
        for k, nm in enumerate("UVd"):
            if UVd[k].numel():
                assert relerr(UVd[k], Uo[k]) <= 2e-4, tag + (nm, relerr(UVd[k], Uo[k]))
        for k in range(3):
            assert relerr(Luvd[k], Lo[k]) <= 2e-4, tag + ("L", k)


LB_CASES = int(os.environ.get("PSGDK_FUZZ_LRA_BF16", "16"))


@pytest.mark.parametrize("seed", list(range(LB_CASES)))
def test_random_lra_bf16_case_matches_oracle(seed):
    """bf16 LRA update + apply with random N (whole 512-row blocks of the packed two-rows-per-thread passes plus a ragged tail, or too short
    for them) and random rank (even ranks 2..16 take the packed passes, odd ones and 18..32 the one-row kernels); both update branches.
    Truth: the fp64 oracle on the same bf16-rounded inputs and draws; yardstick: the oracle in bf16 (torch CPU bf16 arithmetic -- it reproduces
    the reference's bf16 LRAWhiten goldens bit for bit, tests/test_oracle_golden.py): HIP error <= 1.5 x its error + one bf16 ulp."""
    from psgd_torch_amd import lra
    rnd = random.Random(9500 + seed)
    N = rnd.choice([300, 512, 513, 1023, 1024, 1536 + 7, 4096, 5000, 12345, 65536 + 511])
    r = rnd.choice([2, 4, 6, 8, 10, 12, 14, 16, 16, 10, 5, 7, 24])
    bf = torch.bfloat16
    gen = torch.Generator().manual_seed(9600 + seed)
    U = torch.randn(N, r, generator=gen); V = torch.randn(N, r, generator=gen)
    U = (U * (0.1 ** 0.5 / torch.linalg.vector_norm(U))).to(bf); V = (V * (0.1 ** 0.5 / torch.linalg.vector_norm(V))).to(bf)
    d = (0.5 + torch.rand(N, 1, generator=gen)).to(bf)
    UVd = [U.clone().to(DEV).contiguous(), V.clone().to(DEV).contiguous(), d.clone().to(DEV).contiguous()]
    Luvd = [torch.zeros([], device=DEV) for _ in range(3)]
    U64, L64 = [U.double(), V.double(), d.double()], [torch.zeros([], dtype=torch.float64) for _ in range(3)]
    Ub, Lb = [U.clone(), V.clone(), d.clone()], [torch.zeros([], dtype=bf) for _ in range(3)]
    want_packed = (N // 512 * 512) if (r % 2 == 0 and 2 <= r <= 16 and N >= 512) else 0
    ULP = 7.8125e-3
    for t in range(3):
        g = ((0.5 + 2 * torch.rand(N, 1, generator=gen)) * torch.randn(N, 1, generator=gen)).to(bf)
        vn = torch.randn(N, 1, generator=gen).to(bf)
        coin = 0.25 if (t + seed) % 2 == 0 else 0.75
        lra.update_precond_lra_whiten(UVd, Luvd, g.to(DEV), lr=0.1, betaL=0.9, damping=1e-6, v_noise=vn.to(DEV), coin=coin)
        assert UVd[2]._psgdk_lra.info()["packed_rows"] == want_packed, (seed, N, r)
        h = lra.precond_grad_lra(UVd, g.to(DEV))
        orc.update_precond_lra_whiten(U64, L64, g.double(), vn.double(), coin, lr=0.1, betaL=0.9, damping=1e-6)
        h64 = orc.precond_grad_lra(U64, g.double())
        orc.update_precond_lra_whiten(Ub, Lb, g, vn, coin, lr=0.1, betaL=0.9, damping=1e-6)
        hb = orc.precond_grad_lra(Ub, g)
        tag = (seed, N, r, t)
        assert torch.isfinite(h.float()).all(), tag
        for nm, got, ref, truth in (("h", h, hb, h64), ("U", UVd[0], Ub[0], U64[0]), ("V", UVd[1], Ub[1], U64[1]), ("d", UVd[2], Ub[2], U64[2])):
            e_hip, e_ref = relerr(got, truth), relerr(ref, truth)
            assert e_hip <= 1.5 * e_ref + ULP, tag + (nm, e_hip, e_ref)
        for k in range(3):
            e_hip, e_ref = relerr(Luvd[k], L64[k]), relerr(Lb[k], L64[k])
            assert e_hip <= 1.5 * e_ref + 2 * ULP, tag + ("L", k, e_hip, e_ref)
