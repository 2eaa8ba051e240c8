"""
Pins the CPU oracle (oracle/psgd_oracle.py) against golden vectors captured from the reference itself
(tests/golden/gen_golden.py).  CPU only.  Tolerances (relative Frobenius, vs the reference's own output in the
same dtype, identical replayed noise): fp64 1e-10, fp32 5e-6, bf16 4e-2 (bf16 eps = 7.8e-3; the oracle and the
reference differ only in pairwise contraction order, i.e. at the rounding level of each dtype).
"""
import numpy as np
import pytest
import torch

from helpers import DT, P_of, T, golden_names, kron_dtypes, kron_noise_from_golden, load, pro_noise_from_golden, relerr
from oracle import psgd_oracle as orc

TOL = {"fp64": 1e-10, "fp32": 5e-6, "bf16": 4e-2}


def test_golden_fixtures_present():
    assert len(golden_names("kron_")) >= 20
    assert len(golden_names("kroneq_")) >= 8
    assert len(golden_names("kwns4_")) >= 6
    assert len(golden_names("lra_")) >= 3
    assert len(golden_names("lrawhiten_")) >= 2
    assert "helpers" in golden_names("helpers")


@pytest.mark.parametrize("n", [8, 40])
@pytest.mark.parametrize("dn", ["fp64", "fp32", "bf16"])
def test_norm_bounds_and_procrustes(n, dn):
    z = load("helpers")
    dt, key = DT[dn], f"n{n}_{dn}"
    v = orc.norm_lower_bound_spd(T(z[key + "_spd_A"], dt), T(z[key + "_spd_noise"], dt))
    assert relerr(v, z[key + "_spd_out"]) <= TOL[dn]
    v = orc.norm_lower_bound_skh(T(z[key + "_skh_A"], dt), T(z[key + "_skh_noise"], dt))
    assert relerr(v, z[key + "_skh_out"]) <= TOL[dn]
    Q = T(z[key + "_pro_Q"], dt).clone()
    orc.procrustes_step2(Q, T(z[key + "_pro_noise"], dt))
    assert relerr(Q, z[key + "_pro_out"]) <= TOL[dn]


@pytest.mark.parametrize("name", golden_names("kron_"))
def test_kron_update_and_apply(name):
    z = load(name)
    Tn = int(z["T"])
    for dn in kron_dtypes(z):
        dt = DT[dn]
        QL, kinds = orc.init_kron(T(z["G0"], dt), Scale=float(z["Scale"]), max_size=float(z["max_size"]),
                                  max_skew=float(z["max_skew"]))
        for i, q in enumerate(QL[0]):
            assert torch.equal(q.to(torch.float64), T(z[f"{dn}_init_Q{i}"], torch.float64))
            assert (q.dim() == 2) == (kinds[i] == "dense")
        for t in range(Tn):
            G = T(z[f"G{t}"], dt)
            noise = kron_noise_from_golden(z, dn, t, len(QL[0]), dt)
            orc.update_precond_kron_whiten_q0p5eq1p5(QL, G, noise, lr=float(z["lr"]), betaL=float(z["betaL"]),
                                                     damping=float(z["damping"]))
            h = orc.precond_grad_kron(QL[0], G)
            assert relerr(h, z[f"{dn}_t{t}_h"]) <= TOL[dn], (name, dn, t, "h")
            for i, (q, ell) in enumerate(zip(*QL)):
                assert relerr(q, z[f"{dn}_t{t}_Q{i}"]) <= TOL[dn], (name, dn, t, i, "Q")
                assert relerr(ell, z[f"{dn}_t{t}_L{i}"]) <= TOL[dn], (name, dn, t, i, "L")
                assert ell.dtype == (torch.float64 if dn == "fp64" else torch.float32)


@pytest.mark.parametrize("name", golden_names("kroneq_"))
def test_kron_eq_update_and_apply(name):
    """The triangular geometry dQ = E*Q (psgd.py:278-336) against the reference's own outputs."""
    z = load(name)
    Tn = int(z["T"])
    for dn in kron_dtypes(z):
        dt = DT[dn]
        QL, kinds = orc.init_kron(T(z["G0"], dt), Scale=float(z["Scale"]), max_size=float(z["max_size"]),
                                  max_skew=float(z["max_skew"]))
        for t in range(Tn):
            G = T(z[f"G{t}"], dt)
            noise = kron_noise_from_golden(z, dn, t, len(QL[0]), dt)
            orc.update_precond_kron_whiten_eq(QL, G, noise, lr=float(z["lr"]), betaL=float(z["betaL"]),
                                              damping=float(z["damping"]))
            h = orc.precond_grad_kron(QL[0], G)
            assert relerr(h, z[f"{dn}_t{t}_h"]) <= TOL[dn], (name, dn, t, "h")
            for i, (q, ell) in enumerate(zip(*QL)):
                assert relerr(q, z[f"{dn}_t{t}_Q{i}"]) <= TOL[dn], (name, dn, t, i, "Q")
                assert relerr(ell, z[f"{dn}_t{t}_L{i}"]) <= TOL[dn], (name, dn, t, i, "L")
                if q.dim() == 2:
                    assert float(torch.tril(q, -1).abs().max()) == 0.0        # Q stays upper triangular


@pytest.mark.parametrize("name", golden_names("kronqeq_") + golden_names("kronquad_") + golden_names("kronqep_") +
                         golden_names("kronquad4p_"))
def test_kron_qeq_quad_update_and_apply(name):
    """The QEQ / QUAD / QEP / QUAD4P geometries (psgd.py:367-391, 455-483, 339-364, 486-513) against the reference's outputs."""
    fn = {"kronqeq": orc.update_precond_kron_whiten_qeq, "kronquad": orc.update_precond_kron_whiten_quad,
          "kronqep": orc.update_precond_kron_whiten_qep, "kronquad4p": orc.update_precond_kron_whiten_quad4p}[name.split("_")[0]]
    p4 = name.startswith("kronquad4p_")
    z = load(name)
    for dn in kron_dtypes(z):
        dt = DT[dn]
        QL, kinds = orc.init_kron(T(z["G0"], dt), Scale=float(z["Scale"]) ** (2 if p4 else 1), max_size=float(z["max_size"]),
                                  max_skew=float(z["max_skew"]))         # psgd.py:186-187: Scale is squared when fitting P
        for t in range(int(z["T"])):
            G = T(z[f"G{t}"], dt)
            noise = kron_noise_from_golden(z, dn, t, len(QL[0]), dt)
            fn(QL, G, noise, lr=float(z["lr"]), betaL=float(z["betaL"]), damping=float(z["damping"]))
            h = orc.precond_grad_kron_4p(QL[0], G) if p4 else orc.precond_grad_kron(QL[0], G)
            assert relerr(h, z[f"{dn}_t{t}_h"]) <= TOL[dn], (name, dn, t, "h")
            for i, (q, ell) in enumerate(zip(*QL)):
                assert relerr(q, z[f"{dn}_t{t}_Q{i}"]) <= TOL[dn], (name, dn, t, i, "Q")
                assert relerr(ell, z[f"{dn}_t{t}_L{i}"]) <= TOL[dn], (name, dn, t, i, "L")


@pytest.mark.parametrize("name", golden_names("kronpro4p_"))
def test_kron_pro4p_update_and_apply(name):
    """PRO4P (psgd.py:422-452, procrustes_step3 psgd.py:127-158) against the reference's outputs, including the number of
    rotations each dense factor took before its Hermitian-enough break.  fp32 tolerance 1e-3: fitting P directly amplifies
    rounding differences (the reference's own warning, psgd.py:425-426; observed 1.7e-4 between two contraction orders)."""
    tol = {"fp64": 1e-10, "fp32": 1e-3, "bf16": 4e-2}
    z = load(name)
    for dn in kron_dtypes(z):
        dt = DT[dn]
        QL, kinds = orc.init_kron(T(z["G0"], dt), Scale=float(z["Scale"]) ** 2, max_size=float(z["max_size"]),
                                  max_skew=float(z["max_skew"]))
        for t in range(int(z["T"])):
            G = T(z[f"G{t}"], dt)
            noise = kron_noise_from_golden(z, dn, t, len(QL[0]), dt)
            pro = pro_noise_from_golden(z, dn, t, len(QL[0]), dt)
            padded = [None if p is None else p + [torch.zeros_like(p[0])] * (10 - len(p)) for p in pro]
            used = orc.update_precond_kron_whiten_pro4p(QL, G, noise, padded, lr=float(z["lr"]), betaL=float(z["betaL"]),
                                                        damping=float(z["damping"]))
            for i, p in enumerate(pro):
                if p is not None:
                    assert used[i] == len(p), (name, dn, t, i, used[i], len(p))
            h = orc.precond_grad_kron_4p(QL[0], G)
            assert relerr(h, z[f"{dn}_t{t}_h"]) <= tol[dn], (name, dn, t, "h")
            for i, (q, ell) in enumerate(zip(*QL)):
                assert relerr(q, z[f"{dn}_t{t}_Q{i}"]) <= tol[dn], (name, dn, t, i, "Q")
                assert relerr(ell, z[f"{dn}_t{t}_L{i}"]) <= tol[dn], (name, dn, t, i, "L")


def _kw_from_golden(z):
    kw = {}
    for k in z.files:
        if not k.startswith("kw_"):
            continue
        v = z[k]
        name = k[3:]
        if name == "preconditioner_dtype":
            kw[name] = {"none": None, "bf16": torch.bfloat16, "fp32": torch.float32}[str(v)]
        elif name == "dQ":
            kw[name] = str(v)
        elif name == "grad_clip_max_amps":
            kw[name] = tuple(float(x) for x in v)
        elif v.dtype == np.bool_:
            kw[name] = bool(v)
        else:
            kw[name] = float(v)
    return kw


class DrawReplay:
    """Replays the reference's recorded draw stream of one KWNS4.step (SURVEY 8c draw order)."""

    def __init__(self, z, t, dQ="Q0.5EQ1.5"):
        self.z, self.t, self.k = z, t, 0
        self.n = int(z[f"t{t}_ndraws"])
        # only the default geometry rotates (procrustes_step2 -> a second 32 x d draw per dense factor, psgd.py:87); QEP has no
        # balancing gate (it balances on every call, psgd.py:346-347)
        self.has_skh = dQ in ("Q0.5EQ1.5", "Q0p5EQ1p5")
        self.has_gate = dQ != "QEP"

    def _next(self, kind):
        assert self.k < self.n, "oracle consumed more draws than the reference made"
        assert str(self.z[f"t{self.t}_draw{self.k}_kind"]) == kind
        x = self.z[f"t{self.t}_draw{self.k}"]
        self.k += 1
        return x

    def uniform(self):
        return float(self._next("rand"))

    def noise_for(self, G, kinds):
        g_noise = T(self._next("randn"), G.dtype).reshape(G.shape)
        spd, skh = [], []
        for kind in kinds:
            if kind == "dense":
                spd.append(T(self._next("randn"), G.dtype))
                skh.append(T(self._next("randn"), G.dtype) if self.has_skh else None)
            else:
                spd.append(None)
                skh.append(None)
        return orc.KronNoise(g_noise, spd, skh, float(self._next("rand")) if self.has_gate else 1.0)


@pytest.mark.parametrize("name", golden_names("kwns4_"))
def test_kwns4_step(name):
    z = load(name)
    kw = _kw_from_golden(z)
    dQ = str(z["dQ"]) if "dQ" in z.files else "Q0.5EQ1.5"      # fixtures recorded with ..._ddp.py:84-86 switched
    n, Tn = int(z["nparams"]), int(z["T"])
    params = [T(z[f"p{i}_init"], torch.float32).clone() for i in range(n)]
    pd = kw.get("preconditioner_dtype", torch.bfloat16)
    dn = "bf16" if pd == torch.bfloat16 else "fp32"
    replay = {"cur": None}
    opt = orc.KWNS4Oracle(params, uniform=lambda: replay["cur"].uniform(),
                          noise_for=lambda G, kinds: replay["cur"].noise_for(G, kinds), dQ=dQ, **kw)
    for t in range(Tn):
        replay["cur"] = DrawReplay(z, t, dQ)
        grads = [T(z[f"t{t}_g{i}"], torch.float32) for i in range(n)]
        opt.step(grads)
        assert replay["cur"].k == replay["cur"].n, "draw count differs from the reference"
        for i in range(n):
            assert relerr(params[i], z[f"t{t}_p{i}"]) <= (1e-6 if dn == "fp32" else 2e-4), (name, t, i, "p")
            st = opt.state[i]
            if st["ema"] is not None:
                assert relerr(st["ema"], z[f"t{t}_ema{i}"]) <= TOL[dn]
            for j, (q, ell) in enumerate(zip(*st["QL"])):
                assert relerr(q, z[f"t{t}_p{i}_Q{j}"]) <= TOL[dn], (name, t, i, j, "Q")
                assert relerr(ell, z[f"t{t}_p{i}_L{j}"]) <= TOL[dn], (name, t, i, j, "L")


@pytest.mark.parametrize("name", golden_names("lra_") + golden_names("lrabig_"))      # lrabig_: rank 32, oracle only
def test_lra_update_and_apply(name):
    z = load(name)
    Tn = int(z["T"])
    for dn in [d for d in DT if f"{d}_t0_h" in z.files]:
        dt = DT[dn]
        UVd = [T(z["U0"], dt).clone(), T(z["V0"], dt).clone(), T(z["d0"], dt).clone()]
        Luvd = [orc.lift2single(torch.zeros([], dtype=dt)) for _ in range(3)]
        tol = {"fp64": 1e-9, "fp32": 2e-5, "bf16": 6e-2}[dn]
        for t in range(Tn):
            g = T(z[f"g{t}"], dt)
            orc.update_precond_lra_whiten(UVd, Luvd, g, T(z[f"{dn}_t{t}_vnoise"], dt), float(z[f"{dn}_t{t}_coin"]),
                                          lr=float(z["lr"]), betaL=float(z["betaL"]), damping=float(z["damping"]))
            h = orc.precond_grad_lra(UVd, g)
            assert relerr(h, z[f"{dn}_t{t}_h"]) <= tol, (name, dn, t)
            for k, nm in enumerate(("U", "V", "d")):
                assert relerr(UVd[k], z[f"{dn}_t{t}_{nm}"]) <= tol, (name, dn, t, nm)
            for k, nm in enumerate(("Lu", "Lv", "Ld")):
                assert relerr(Luvd[k], z[f"{dn}_t{t}_{nm}"]) <= tol, (name, dn, t, nm)


@pytest.mark.parametrize("name", golden_names("lrawhiten_"))
def test_lrawhiten_step(name):
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
                continue
            else:
                kw[nm] = None if np.isnan(float(v)) else float(v)
    # (lrawhiten_bf16_*: bf16 parameters -- the reference keeps U, V, d and the momentum in the parameter dtype; the oracle in bf16 against the
    #  reference in bf16: torch's CPU bf16 arithmetic on both sides, differences only where the contraction order differs)
    dt = torch.bfloat16 if name.startswith("lrawhiten_bf16") else torch.float32
    tol_p, tol_q = (1e-6, 2e-5) if dt == torch.float32 else (1e-3, 5e-3)      # (measured here: 0.0 -- the same bits)
    params = [T(z[f"p{i}_init"], dt).clone() for i in range(3)]
    opt = orc.LRAWhitenOracle(params, T(z["U0"], dt), T(z["V0"], dt), **kw)
    for t in range(Tn):
        grads = [T(z[f"t{t}_g{i}"], dt) for i in range(3)]
        nd = int(z[f"t{t}_ndraws"])
        kinds = [str(z[f"t{t}_draw{k}_kind"]) for k in range(nd)]
        gate_u = float(z[f"t{t}_draw0"])
        assert kinds[0] == "rand"
        v_noise = coin = None
        if nd > 1:
            assert kinds[1:] == ["randn", "rand"]
            v_noise, coin = T(z[f"t{t}_draw1"], dt), float(z[f"t{t}_draw2"])
        opt.step(grads, gate_u, v_noise, coin)
        for i in range(3):
            assert relerr(params[i], z[f"t{t}_p{i}"]) <= tol_p, (name, t, i, relerr(params[i], z[f"t{t}_p{i}"]))
        for k, nm in enumerate(("U", "V", "d")):
            assert relerr(opt.UVd[k], z[f"t{t}_{nm}"]) <= tol_q, (name, t, nm, relerr(opt.UVd[k], z[f"t{t}_{nm}"]))


def kronwhiten_kw_from_golden(z):
    kw = {}
    for k in z.files:
        if not k.startswith("kw_"):
            continue
        nm, v = k[3:], z[k]
        if nm == "dQ":
            kw[nm] = str(v)
        elif nm == "grad_clip_max_amps":
            kw[nm] = tuple(float(x) for x in v)
        elif v.dtype == np.bool_:
            kw[nm] = bool(v)
        else:
            kw[nm] = None if np.isnan(float(v)) else float(v)
    return kw


class KronWhitenReplay(DrawReplay):
    """The recorded draw stream of one KronWhiten.step (psgd.py:589-654): the gate, then per updated tensor randn_like(G),
    per dense factor the spd draw (+ the skh draw for Q0.5EQ1.5; QUAD4P has no Procrustes step), the balancing rand([])."""

    def __init__(self, z, t, p4):
        super().__init__(z, t)
        self.p4 = p4

    def noise_for(self, G, kinds):
        g_noise = T(self._next("randn"), G.dtype).reshape(G.shape)
        spd, skh = [], []
        for kind in kinds:
            if kind == "dense":
                spd.append(T(self._next("randn"), G.dtype))
                skh.append(None if self.p4 else T(self._next("randn"), G.dtype))
            else:
                spd.append(None)
                skh.append(None)
        return orc.KronNoise(g_noise, spd, skh, float(self._next("rand")))


@pytest.mark.parametrize("name", golden_names("kronwhiten_"))
def test_kronwhiten_step(name):
    """KronWhitenOracle (psgd.py:589-654 restated) against fixtures captured from psgd.KronWhiten.step itself: on-the-fly
    initial scale, all-updates-then-all-applies draw order, per-tensor clipping, momentum on/off, gated updates, QUAD4P."""
    z = load(name)
    kw = kronwhiten_kw_from_golden(z)
    n, Tn = int(z["nparams"]), int(z["T"])
    params = [T(z[f"p{i}_init"], torch.float32).clone() for i in range(n)]
    replay = {"cur": None}
    opt = orc.KronWhitenOracle(params, uniform=lambda: replay["cur"].uniform(),
                               noise_for=lambda G, kinds: replay["cur"].noise_for(G, kinds), **kw)
    for t in range(Tn):
        replay["cur"] = KronWhitenReplay(z, t, kw.get("dQ") == "QUAD4P")
        opt.step([T(z[f"t{t}_g{i}"], torch.float32) for i in range(n)])
        assert replay["cur"].k == replay["cur"].n, "draw count differs from the reference"
        for i in range(n):
            assert relerr(params[i], z[f"t{t}_p{i}"]) <= 1e-6, (name, t, i, "p")
            if opt.ms is not None:
                assert relerr(opt.ms[i], z[f"t{t}_m{i}"]) <= TOL["fp32"]
            for j, (q, ell) in enumerate(zip(*opt.QLs[i])):
                assert relerr(q, z[f"t{t}_p{i}_Q{j}"]) <= TOL["fp32"], (name, t, i, j, "Q")
                assert relerr(ell, z[f"t{t}_p{i}_L{j}"]) <= TOL["fp32"], (name, t, i, j, "L")
