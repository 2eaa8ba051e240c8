#!/usr/bin/env python3
"""
Golden-vector generator.  Runs ONLY in the build container (needs /root/reference); the GPU box and the
test-suite never execute it -- they read the committed tests/golden/*.npz it wrote.

What it does
  * imports the reference's psgd.py and wrapped_as_torch_optimizer_for_ddp.py from /root/reference;
  * the reference imports `opt_einsum` unconditionally (psgd.py:42), which is not installed in this image and
    is not pinned by the reference (no requirements/lock file).  opt_einsum contributes only the ORDER of the
    pairwise contractions (the arithmetic is torch matmul), so an in-process stand-in is installed under that
    module name: `get_symbol` (same symbol table) and `contract_expression` (numpy.einsum_path order, pairwise
    torch.einsum).  Parity is therefore pinned to the reference's own Python arithmetic with contraction order
    unpinned -- exactly the state of the upstream project;
  * wraps torch.randn / randn_like / rand so that every random draw of the reference is RECORDED (and, where a
    fixture wants to force a rare branch such as the 1%-probability balancing, overridden), then stores
    inputs + recorded draws + outputs as small .npz fixtures.

Fixtures are data only (inputs and expected outputs); no reference source text is stored.
Usage:  python tests/golden/gen_golden.py            (rewrites tests/golden/*.npz)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


# ------------------------------------------------------------------------------------------------
# opt_einsum stand-in (contraction ORDER only)
# ------------------------------------------------------------------------------------------------
def _install_opt_einsum_standin():
    mod = types.ModuleType("opt_einsum")
    base = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"

    def get_symbol(i):
        if i < 52:
            return base[i]
        return chr(i + 140)

    class _Expr:
        def __init__(self, subscripts, *shapes):
            lhs, out = subscripts.split("->")
            terms = lhs.split(",")
            symbols = []
            for t in terms + [out]:
                for c in t:
                    if c not in symbols:
                        symbols.append(c)
            assert len(symbols) <= 52
            remap = {c: base[k] for k, c in enumerate(symbols)}
            self.terms = ["".join(remap[c] for c in t) for t in terms]
            self.out = "".join(remap[c] for c in out)
            dummies = [np.empty(tuple(s)) for s in shapes]
            mode = "optimal" if len(terms) <= 6 else "greedy"
            path, _ = np.einsum_path(",".join(self.terms) + "->" + self.out, *dummies, optimize=mode)
            self.path = [p for p in path[1:]]

        def __call__(self, *ops):
            ops = list(ops)
            terms = list(self.terms)
            for contract in self.path:
                idx = sorted(contract, reverse=True)
                sel_ops = [ops.pop(i) for i in idx]
                sel_terms = [terms.pop(i) for i in idx]
                remaining = set("".join(terms)) | set(self.out)
                if len(terms) == 0:
                    new_term = self.out
                else:
                    seen = []
                    for t in sel_terms:
                        for c in t:
                            if c in remaining and c not in seen:
                                seen.append(c)
                    new_term = "".join(seen)
                res = torch.einsum(",".join(sel_terms) + "->" + new_term, *sel_ops)
                ops.append(res)
                terms.append(new_term)
            assert len(ops) == 1
            return ops[0]

    def contract_expression(subscripts, *shapes, **kw):
        return _Expr(subscripts, *shapes)

    def contract(subscripts, *ops, **kw):
        return _Expr(subscripts, *[o.shape for o in ops])(*ops)

    mod.get_symbol = get_symbol
    mod.contract_expression = contract_expression
    mod.contract = contract
    sys.modules["opt_einsum"] = mod


_install_opt_einsum_standin()
torch.backends.opt_einsum.enabled = False
sys.path.insert(0, REF)
import psgd  # noqa: E402  (the reference)
import wrapped_as_torch_optimizer_for_ddp as ref_ddp  # noqa: E402


# ------------------------------------------------------------------------------------------------
# RNG recorder
# ------------------------------------------------------------------------------------------------
class Recorder:
    """Wraps torch.randn / randn_like / rand; records each draw in call order.  `force_rand` is a list of
    values (or None) consumed by successive torch.rand([]) calls to force rare branches."""

    def __init__(self, force_rand=None):
        self.draws = []
        self.force_rand = list(force_rand) if force_rand else []
        self._orig = (torch.randn, torch.randn_like, torch.rand)

    def __enter__(self):
        o_randn, o_randn_like, o_rand = self._orig

        def randn(*a, **k):
            x = o_randn(*a, **k)
            self.draws.append(("randn", x.clone()))
            return x

        def randn_like(t, **k):
            x = o_randn_like(t, **k)
            self.draws.append(("randn", x.clone()))
            return x

        def rand(*a, **k):
            x = o_rand(*a, **k)
            if x.dim() == 0 and self.force_rand:
                f = self.force_rand.pop(0)
                if f is not None:
                    x = torch.tensor(f, dtype=x.dtype)
            self.draws.append(("rand", x.clone()))
            return x

        torch.randn, torch.randn_like, torch.rand = randn, randn_like, rand
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like, torch.rand = self._orig
        return False


DT = {"fp64": torch.float64, "fp32": torch.float32, "bf16": torch.bfloat16}


def npy(x):
    """Store bf16/fp32 as float32 (lossless), fp64 as float64."""
    if isinstance(x, (float, int)):
        return np.asarray(x, dtype=np.float64)
    if x.dtype == torch.float64:
        return x.detach().numpy().copy()
    return x.detach().to(torch.float32).numpy().copy()


def save(name, d):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d)
    print(f"{name}.npz  {os.path.getsize(path)/1024:.1f} KiB  ({len(d)} arrays)")


# ------------------------------------------------------------------------------------------------
# A. spectral-norm lower bounds + Procrustes step
# ------------------------------------------------------------------------------------------------
def gen_helpers():
    out = {}
    g = torch.Generator().manual_seed(11)
    for n in (8, 40):
        X = torch.randn(n, 3 * n, generator=g, dtype=torch.float64)
        spd64 = X @ X.t() / n
        W = torch.randn(n, n, generator=g, dtype=torch.float64)
        skh64 = 0.3 * (W - W.t())
        Q64 = torch.eye(n, dtype=torch.float64) + 0.1 * torch.randn(n, n, generator=g, dtype=torch.float64)
        for dn, dt in DT.items():
            A = spd64.to(dt)
            A = (A + A.t()) / 2 if dt != torch.bfloat16 else torch.triu(A) + torch.triu(A, 1).t()
            S = skh64.to(dt)
            S = torch.triu(S, 1) - torch.triu(S, 1).t()
            Q = Q64.to(dt)
            key = f"n{n}_{dn}"
            torch.manual_seed(100 + n)
            with Recorder() as r:
                val = psgd.norm_lower_bound_spd(A)
            out[key + "_spd_A"], out[key + "_spd_noise"], out[key + "_spd_out"] = npy(A), npy(r.draws[0][1]), npy(val)
            with Recorder() as r:
                val = psgd.norm_lower_bound_skh(S)
            out[key + "_skh_A"], out[key + "_skh_noise"], out[key + "_skh_out"] = npy(S), npy(r.draws[0][1]), npy(val)
            Qc = Q.clone()
            with Recorder() as r:
                psgd.procrustes_step2(Qc)
            out[key + "_pro_Q"], out[key + "_pro_noise"], out[key + "_pro_out"] = npy(Q), npy(r.draws[0][1]), npy(Qc)
    save("helpers", out)


# ------------------------------------------------------------------------------------------------
# B. functional seam: init_kron / update_precond_kron_whiten_q0p5eq1p5 / precond_grad_kron
# ------------------------------------------------------------------------------------------------
def structured_grads(shape, T, seed):
    """fp32 gradient stream G_t = H_1 x_1 ... (mode-wise SPD mixing) so that Q actually moves."""
    g = torch.Generator().manual_seed(seed)
    if len(shape) == 0:
        return [torch.randn([], generator=g) * 0.7 for _ in range(T)]
    mixers = []
    for s in shape:
        W = torch.randn(s, s, generator=g) / (s ** 0.5)
        mixers.append(torch.eye(s) * 0.5 + W @ W.t())
    out = []
    for _ in range(T):
        X = torch.randn(*shape, generator=g)
        for i, M in enumerate(mixers):
            X = torch.movedim(torch.tensordot(M, torch.movedim(X, i, 0), dims=1), 0, i)
        out.append(0.3 * X)
    return out


def gen_kron_case(name, shape, dtypes, T, max_skew=1.0, max_size=float("inf"), Scale=1.0, lr=0.5, betaL=0.9,
                  damping=1e-9, force_balance_at=None, seed=0):
    out = {"shape": np.asarray(shape, dtype=np.int64), "T": np.asarray(T), "max_skew": np.asarray(max_skew),
           "max_size": np.asarray(max_size), "Scale": np.asarray(Scale), "lr": np.asarray(lr),
           "betaL": np.asarray(betaL), "damping": np.asarray(damping)}
    G32 = structured_grads(shape, T, seed + 1)
    for t in range(T):
        out[f"G{t}"] = npy(G32[t])
    for dn in dtypes:
        dt = DT[dn]
        QL, exprs = psgd.init_kron(G32[0].to(dt), Scale=Scale, max_size=max_size, max_skew=max_skew, dQ="Q0.5EQ1.5")
        for i, q in enumerate(QL[0]):
            out[f"{dn}_init_Q{i}"] = npy(q)
        torch.manual_seed(1000 + seed)
        for t in range(T):
            G = G32[t].to(dt)
            ndense = sum(1 for q in QL[0] if q.dim() == 2)
            force = [0.001 if force_balance_at == t else 0.5]
            with Recorder(force_rand=force) as r:
                psgd.update_precond_kron_whiten_q0p5eq1p5(QL, exprs, G, lr=lr, betaL=betaL, damping=damping)
            assert len(r.draws) == 1 + 2 * ndense + 1, (len(r.draws), ndense)
            out[f"{dn}_t{t}_gnoise"] = npy(r.draws[0][1])
            k = 1
            for i, q in enumerate(QL[0]):
                if q.dim() == 2:
                    out[f"{dn}_t{t}_spd{i}"] = npy(r.draws[k][1])
                    out[f"{dn}_t{t}_skh{i}"] = npy(r.draws[k + 1][1])
                    k += 2
            out[f"{dn}_t{t}_balance_u"] = npy(r.draws[k][1])
            h = psgd.precond_grad_kron(QL, exprs, G)
            out[f"{dn}_t{t}_h"] = npy(h)
            for i, (q, ell) in enumerate(zip(*QL)):
                out[f"{dn}_t{t}_Q{i}"] = npy(q)
                out[f"{dn}_t{t}_L{i}"] = npy(ell)
    save("kron_" + name, out)


LENET5 = [(6, 26), (16, 151), (257, 120), (121, 84), (85, 10)]


def gen_kron():
    all3 = ("fp64", "fp32", "bf16")
    gen_kron_case("scalar", (), all3, T=4, seed=1)
    gen_kron_case("vec33", (33,), all3, T=4, seed=2)
    gen_kron_case("m48x32", (48, 32), all3, T=10, seed=3)                      # [diag, dense]
    gen_kron_case("m32x48", (32, 48), all3, T=6, seed=4)                       # [dense, diag]
    gen_kron_case("m64x64", (64, 64), all3, T=8, seed=5, force_balance_at=3)  # [dense, dense] + balance branch
    gen_kron_case("m40x8", (40, 8), all3, T=4, seed=6)
    gen_kron_case("m24x40_diagdiag", (24, 40), all3, T=4, max_skew=0.0, seed=7)   # [diag, diag]
    gen_kron_case("m40x24_maxsize", (40, 24), all3, T=4, max_skew=float("inf"), max_size=30, seed=8)
    gen_kron_case("t7x5x3", (7, 5, 3), all3, T=5, seed=9, force_balance_at=2)  # 3 dense factors
    gen_kron_case("t4x6x5x3_mixed", (4, 6, 5, 3), ("fp64", "fp32"), T=3, max_skew=0.2, seed=10)
    gen_kron_case("m768x96", (768, 96), ("fp32", "bf16"), T=2, seed=11)
    for k, shp in enumerate(LENET5):
        gen_kron_case(f"lenet{k}_skew1", shp, ("fp32", "bf16"), T=2, seed=20 + k)
        gen_kron_case(f"lenet{k}_skewinf", shp, ("fp32",), T=2, max_skew=float("inf"), seed=30 + k)
    gen_kron_case("scale_init", (20, 12), ("fp64", "fp32"), T=3, max_skew=float("inf"), Scale=0.1, lr=0.3,
                  betaL=0.5, damping=1e-3, seed=40)


def gen_kron_eq_case(name, shape, dtypes, T, max_skew=1.0, max_size=float("inf"), Scale=1.0, lr=0.2, betaL=0.9,
                     damping=1e-9, force_balance_at=None, seed=0):
    """The triangular geometry dQ = E*Q (psgd.py:278-336): init_kron(dQ="EQ") + update_precond_kron_whiten_eq."""
    out = {"shape": np.asarray(shape, dtype=np.int64), "T": np.asarray(T), "max_skew": np.asarray(max_skew),
           "max_size": np.asarray(max_size), "Scale": np.asarray(Scale), "lr": np.asarray(lr),
           "betaL": np.asarray(betaL), "damping": np.asarray(damping)}
    G32 = structured_grads(shape, T, seed + 1)
    for t in range(T):
        out[f"G{t}"] = npy(G32[t])
    for dn in dtypes:
        dt = DT[dn]
        QL, exprs = psgd.init_kron(G32[0].to(dt), Scale=Scale, max_size=max_size, max_skew=max_skew, dQ="EQ")
        torch.manual_seed(2000 + seed)
        for t in range(T):
            G = G32[t].to(dt)
            ndense = sum(1 for q in QL[0] if q.dim() == 2)
            force = [0.001 if force_balance_at == t else 0.5]
            with Recorder(force_rand=force) as r:
                psgd.update_precond_kron_whiten_eq(QL, exprs, G, lr=lr, betaL=betaL, damping=damping)
            assert len(r.draws) == 1 + ndense + 1, (len(r.draws), ndense)
            out[f"{dn}_t{t}_gnoise"] = npy(r.draws[0][1])
            k = 1
            for i, q in enumerate(QL[0]):
                if q.dim() == 2:
                    out[f"{dn}_t{t}_spd{i}"] = npy(r.draws[k][1])
                    k += 1
            out[f"{dn}_t{t}_balance_u"] = npy(r.draws[k][1])
            out[f"{dn}_t{t}_h"] = npy(psgd.precond_grad_kron(QL, exprs, G))
            for i, (q, ell) in enumerate(zip(*QL)):
                out[f"{dn}_t{t}_Q{i}"] = npy(q)
                out[f"{dn}_t{t}_L{i}"] = npy(ell)
    save("kroneq_" + name, out)


def gen_kron_geom_case(geom, name, shape, dtypes, T, max_skew=1.0, max_size=float("inf"), Scale=1.0, lr=0.3, betaL=0.9,
                       damping=1e-9, force_balance_at=None, seed=0):
    """The QEQ / QUAD geometries (psgd.py:367-391, 455-483) behind the same seam."""
    fn = {"QEQ": psgd.update_precond_kron_whiten_qeq, "QUAD": psgd.update_precond_kron_whiten_quad,
          "QEP": psgd.update_precond_kron_whiten_qep, "QUAD4P": psgd.update_precond_kron_whiten_quad4p}[geom]
    out = {"shape": np.asarray(shape, dtype=np.int64), "T": np.asarray(T), "max_skew": np.asarray(max_skew),
           "max_size": np.asarray(max_size), "Scale": np.asarray(Scale), "lr": np.asarray(lr),
           "betaL": np.asarray(betaL), "damping": np.asarray(damping)}
    G32 = structured_grads(shape, T, seed + 1)
    for t in range(T):
        out[f"G{t}"] = npy(G32[t])
    for dn in dtypes:
        dt = DT[dn]
        QL, exprs = psgd.init_kron(G32[0].to(dt), Scale=Scale, max_size=max_size, max_skew=max_skew, dQ=geom)
        torch.manual_seed(3000 + seed)
        for t in range(T):
            G = G32[t].to(dt)
            ndense = sum(1 for q in QL[0] if q.dim() == 2)
            force = [0.001 if force_balance_at == t else 0.5]
            with Recorder(force_rand=force) as r:
                fn(QL, exprs, G, lr=lr, betaL=betaL, damping=damping)
            assert len(r.draws) == 1 + ndense + (0 if geom == "QEP" else 1), (len(r.draws), ndense)
            out[f"{dn}_t{t}_gnoise"] = npy(r.draws[0][1])
            k = 1
            for i, q in enumerate(QL[0]):
                if q.dim() == 2:
                    out[f"{dn}_t{t}_spd{i}"] = npy(r.draws[k][1])
                    k += 1
            out[f"{dn}_t{t}_balance_u"] = npy(r.draws[k][1]) if geom != "QEP" else np.asarray(0.5)   # QEP draws no gate
            if geom == "QUAD4P":       # KronWhiten applies exprA(*Q, G) for the P-fitting geometries (psgd.py:573)
                out[f"{dn}_t{t}_h"] = npy(exprs[0](*QL[0], G))
            else:
                out[f"{dn}_t{t}_h"] = npy(psgd.precond_grad_kron(QL, exprs, G))
            for i, (q, ell) in enumerate(zip(*QL)):
                out[f"{dn}_t{t}_Q{i}"] = npy(q)
                out[f"{dn}_t{t}_L{i}"] = npy(ell)
    save(f"kron{geom.lower()}_" + name, out)


def gen_kron_geoms():
    all3 = ("fp64", "fp32", "bf16")
    for k, geom in enumerate(("QEQ", "QUAD", "QEP")):
        b = 70 + 10 * k
        gen_kron_geom_case(geom, "vec33", (33,), all3, T=4, seed=b + 1)
        gen_kron_geom_case(geom, "m48x32", (48, 32), all3, T=8, seed=b + 2)
        gen_kron_geom_case(geom, "m32x48", (32, 48), all3, T=6, seed=b + 3)
        gen_kron_geom_case(geom, "m64x64", (64, 64), all3, T=4, seed=b + 4, force_balance_at=2)
        gen_kron_geom_case(geom, "m150x200", (150, 200), ("fp32", "bf16"), T=2, seed=b + 5)
        gen_kron_geom_case(geom, "t7x5x3", (7, 5, 3), ("fp64", "fp32"), T=3, seed=b + 6)
    # QUAD4P (psgd.py:486-513) fits P itself: own recipe (Scale 0.8 so that P = Q starts away from the identity)
    kw = dict(Scale=0.8)
    gen_kron_geom_case("QUAD4P", "vec33", (33,), all3, T=4, seed=111, **kw)
    gen_kron_geom_case("QUAD4P", "m48x32", (48, 32), all3, T=6, seed=112, **kw)
    gen_kron_geom_case("QUAD4P", "m32x48", (32, 48), all3, T=5, seed=113, **kw)
    gen_kron_geom_case("QUAD4P", "m64x64", (64, 64), all3, T=4, seed=114, force_balance_at=2, **kw)
    gen_kron_geom_case("QUAD4P", "m150x200", (150, 200), ("fp32", "bf16"), T=2, seed=115, **kw)
    gen_kron_geom_case("QUAD4P", "t7x5x3", (7, 5, 3), ("fp64", "fp32"), T=3, seed=116, **kw)


def gen_kron_pro4p_case(name, shape, dtypes, T, max_skew=1.0, max_size=float("inf"), Scale=0.8, lr=0.3, betaL=0.9,
                        damping=1e-9, force_balance_at=None, seed=0):
    """PRO4P (psgd.py:422-452): fits P directly; a variable number (1..10) of procrustes_step3 calls per dense factor, each
    with its own (32, d) draw.  Stored per step and dense factor i: spd{i}, npro{i}, pro{i}_{k}."""
    out = {"shape": np.asarray(shape, dtype=np.int64), "T": np.asarray(T), "max_skew": np.asarray(max_skew),
           "max_size": np.asarray(max_size), "Scale": np.asarray(Scale), "lr": np.asarray(lr),
           "betaL": np.asarray(betaL), "damping": np.asarray(damping)}
    G32 = structured_grads(shape, T, seed + 1)
    for t in range(T):
        out[f"G{t}"] = npy(G32[t])
    events = []
    o_spd, o_pro = psgd.norm_lower_bound_spd, psgd.procrustes_step3

    def spd(*a, **k):
        events.append("spd")
        return o_spd(*a, **k)

    def pro(*a, **k):
        events.append("pro")
        return o_pro(*a, **k)
    psgd.norm_lower_bound_spd, psgd.procrustes_step3 = spd, pro
    try:
        for dn in dtypes:
            dt = DT[dn]
            QL, exprs = psgd.init_kron(G32[0].to(dt), Scale=Scale, max_size=max_size, max_skew=max_skew, dQ="PRO4P")
            torch.manual_seed(4000 + seed)
            for t in range(T):
                G = G32[t].to(dt)
                del events[:]
                force = [0.001 if force_balance_at == t else 0.5]
                with Recorder(force_rand=force) as r:
                    psgd.update_precond_kron_whiten_pro4p(QL, exprs, G, lr=lr, betaL=betaL, damping=damping)
                counts = []
                for e in events:
                    if e == "spd":
                        counts.append(0)
                    else:
                        counts[-1] += 1
                out[f"{dn}_t{t}_gnoise"] = npy(r.draws[0][1])
                k, c = 1, 0
                for i, q in enumerate(QL[0]):
                    if q.dim() == 2:
                        out[f"{dn}_t{t}_spd{i}"] = npy(r.draws[k][1]); k += 1
                        out[f"{dn}_t{t}_npro{i}"] = np.asarray(counts[c])
                        for j in range(counts[c]):
                            out[f"{dn}_t{t}_pro{i}_{j}"] = npy(r.draws[k][1]); k += 1
                        c += 1
                out[f"{dn}_t{t}_balance_u"] = npy(r.draws[k][1])
                assert k + 1 == len(r.draws), (k, len(r.draws))
                out[f"{dn}_t{t}_h"] = npy(exprs[0](*QL[0], G))
                for i, (q, ell) in enumerate(zip(*QL)):
                    out[f"{dn}_t{t}_Q{i}"] = npy(q)
                    out[f"{dn}_t{t}_L{i}"] = npy(ell)
    finally:
        psgd.norm_lower_bound_spd, psgd.procrustes_step3 = o_spd, o_pro
    save("kronpro4p_" + name, out)


def gen_kron_pro4p():
    all3 = ("fp64", "fp32", "bf16")
    gen_kron_pro4p_case("vec33", (33,), all3, T=4, seed=131)
    gen_kron_pro4p_case("m48x32", (48, 32), all3, T=6, seed=132)
    gen_kron_pro4p_case("m32x48", (32, 48), all3, T=5, seed=133)
    gen_kron_pro4p_case("m64x64", (64, 64), all3, T=4, seed=134, force_balance_at=2, lr=0.6)
    gen_kron_pro4p_case("m150x200", (150, 200), ("fp32", "bf16"), T=2, seed=135)
    gen_kron_pro4p_case("t7x5x3", (7, 5, 3), ("fp64", "fp32"), T=3, seed=136, lr=0.6)


def gen_kron_eq():
    all3 = ("fp64", "fp32", "bf16")
    gen_kron_eq_case("scalar", (), all3, T=4, seed=51)
    gen_kron_eq_case("vec33", (33,), all3, T=4, seed=52)
    gen_kron_eq_case("m48x32", (48, 32), all3, T=8, seed=53)                      # [diag, dense]
    gen_kron_eq_case("m32x48", (32, 48), all3, T=6, seed=54)                      # [dense, diag]
    gen_kron_eq_case("m64x64", (64, 64), all3, T=4, seed=55, force_balance_at=2)  # [dense, dense] + balance branch
    gen_kron_eq_case("m24x40_diagdiag", (24, 40), all3, T=4, max_skew=0.0, seed=57)
    gen_kron_eq_case("m150x200", (150, 200), ("fp32", "bf16"), T=2, seed=58)      # several 64-blocks per triangular solve
    gen_kron_eq_case("m257x120", (257, 120), ("fp32", "bf16"), T=2, max_skew=float("inf"), seed=59)
    gen_kron_eq_case("t7x5x3", (7, 5, 3), ("fp64", "fp32"), T=3, seed=60)          # oracle only: N-D under EQ is not built


# ------------------------------------------------------------------------------------------------
# C. KWNS4.step (the torch.optim shell the build mirrors)
# ------------------------------------------------------------------------------------------------
KW_SHAPES = [(48, 32), (32,), (1, 16, 1), (24, 24), (1,), (20, 30)]


def gen_kwns4_case(name, T=4, grad_scale=0.3, seed=0, force_gate=None, dQ=None, **kw):
    """dQ: the reference's own switch -- "one can change these 3 lines to switch the preconditioner"
    (wrapped_as_torch_optimizer_for_ddp.py:84-86) -- applied to the constructed optimizer the way psgd.KronWhiten makes the
    same choice (psgd.py:565-586)."""
    out = {"T": np.asarray(T), "nparams": np.asarray(len(KW_SHAPES))}
    if dQ is not None:
        out["dQ"] = np.asarray(dQ)
    for k, v in kw.items():
        if k == "preconditioner_dtype":
            out["kw_" + k] = np.asarray({None: "none", torch.bfloat16: "bf16", torch.float32: "fp32"}[v])
        elif k == "grad_clip_max_amps":
            out["kw_" + k] = np.asarray(v, dtype=np.float64)
        else:
            out["kw_" + k] = np.asarray(v)
    g = torch.Generator().manual_seed(500 + seed)
    params = [torch.nn.Parameter(0.5 * torch.randn(*s, generator=g)) for s in KW_SHAPES]
    for i, p in enumerate(params):
        out[f"p{i}_init"] = npy(p.data)
        out[f"p{i}_shape"] = np.asarray(p.shape, dtype=np.int64)
    opt = ref_ddp.KWNS4(params, **kw)
    assert not opt.is_distributed
    if dQ is not None:
        opt.dQ = dQ
        opt.update_precond = {"EQ": psgd.update_precond_kron_whiten_eq, "QEQ": psgd.update_precond_kron_whiten_qeq,
                              "QUAD": psgd.update_precond_kron_whiten_quad, "QEP": psgd.update_precond_kron_whiten_qep,
                              "QUAD4P": psgd.update_precond_kron_whiten_quad4p}[dQ]
        if dQ == "QUAD4P":
            opt.precond_grad = lambda QL, exprs, G: exprs[0](*QL[0], G)
    torch.manual_seed(900 + seed)
    for t in range(T):
        grads = []
        for i, s in enumerate(KW_SHAPES):
            sq = tuple(d for d in s if d != 1)
            G = structured_grads(sq, 1, seed * 1000 + 17 * t + i)[0].reshape(s) * (grad_scale / 0.3)
            grads.append(G)
            params[i].grad = G.clone()
            out[f"t{t}_g{i}"] = npy(G)
        force = [force_gate[t]] if force_gate is not None else None
        with Recorder(force_rand=force) as r:
            opt.step()
        # split the recorded draw stream: first draw = group gate; then per tensor, per update call:
        # randn_like(G), [randn(32,d) x2 per dense factor], rand([])
        out[f"t{t}_ndraws"] = np.asarray(len(r.draws))
        for k, (kind, x) in enumerate(r.draws):
            out[f"t{t}_draw{k}"] = npy(x)
            out[f"t{t}_draw{k}_kind"] = np.asarray(kind)
        for i, p in enumerate(params):
            out[f"t{t}_p{i}"] = npy(p.data)
            st = opt.state[p]
            if st["ema"] is not None:
                out[f"t{t}_ema{i}"] = npy(st["ema"])
            for j, (q, ell) in enumerate(zip(*st["QL"])):
                out[f"t{t}_p{i}_Q{j}"] = npy(q)
                out[f"t{t}_p{i}_L{j}"] = npy(ell)
    save("kwns4_" + name, out)


def gen_kwns4():
    gen_kwns4_case("default_bf16", seed=1)
    gen_kwns4_case("fp32_whitengrad_last_coupled", seed=2, preconditioner_dtype=torch.float32, whiten_grad=True,
                   update_preconditioner_first=False, decoupled_weight_decay=False, weight_decay=0.01)
    gen_kwns4_case("fp32_nomomentum", seed=3, preconditioner_dtype=torch.float32, whiten_grad=True, momentum=0.0,
                   weight_decay=0.0)
    gen_kwns4_case("bf16_prob_skewinf", seed=4, T=6, preconditioner_update_probability=0.5,
                   preconditioner_max_skew=float("inf"), force_gate=[0.1, 0.9, 0.2, 0.8, 0.3, 0.7])
    gen_kwns4_case("none_dtype_clip", seed=5, preconditioner_dtype=None, preconditioner_init_scale=10.0,
                   grad_scale=2.0, lr_params=1e-3, grad_clip_max_amps=(1.5, 3.0))
    # the geometry switch of ..._ddp.py:84-86 (fp32: the factor Q itself is compared for these geometries)
    gen_kwns4_case("dq_quad_fp32", seed=7, dQ="QUAD", preconditioner_dtype=torch.float32, lr_preconditioner=0.3)
    gen_kwns4_case("dq_eq_fp32_last", seed=8, dQ="EQ", preconditioner_dtype=torch.float32, lr_preconditioner=0.2,
                   update_preconditioner_first=False, whiten_grad=True)
    gen_kwns4_case("dq_qeq_bf16", seed=9, dQ="QEQ", lr_preconditioner=0.3)
    gen_kwns4_case("dq_quad4p_fp32", seed=10, dQ="QUAD4P", preconditioner_dtype=torch.float32, lr_preconditioner=0.3,
                   preconditioner_init_scale=0.7)
    gen_kwns4_case("dq_qep_fp32", seed=11, dQ="QEP", preconditioner_dtype=torch.float32, lr_preconditioner=0.3)
    gen_kwns4_case("fp32_maxsize", seed=6, preconditioner_dtype=torch.float32, preconditioner_max_size=25,
                   preconditioner_max_skew=float("inf"), lr_preconditioner=0.2, betaL=0.8, damping=1e-4)


# ------------------------------------------------------------------------------------------------
# C2. KronWhiten.step (psgd.py:589-654): the closure-style shell -- on-the-fly initial scale (:599-602), all updates then all
#     applies (:620-639), per-tensor clipping (:642-651)
# ------------------------------------------------------------------------------------------------
KWH_SHAPES = [(24, 16), (16,), (3, 4, 5), (1, 12, 1), (20, 20), ()]


def gen_kronwhiten_case(name, T=4, seed=0, grad_scale=1.0, force_gate=None, **kw):
    out = {"T": np.asarray(T), "nparams": np.asarray(len(KWH_SHAPES))}
    for k, v in kw.items():
        if k == "grad_clip_max_amps":
            out["kw_" + k] = np.asarray(v, dtype=np.float64)
        elif k == "dQ":
            out["kw_" + k] = np.asarray(v)
        else:
            out["kw_" + k] = np.asarray(v if v is not None else float("nan"))
    g = torch.Generator().manual_seed(600 + seed)
    params = [torch.nn.Parameter(0.5 * torch.randn(s, generator=g)) for s in KWH_SHAPES]
    for i, p in enumerate(params):
        out[f"p{i}_init"] = npy(p.data)
        out[f"p{i}_shape"] = np.asarray(p.shape, dtype=np.int64)
    opt = psgd.KronWhiten(params, **kw)
    torch.manual_seed(950 + seed)
    for t in range(T):
        cs = []
        for i, s in enumerate(KWH_SHAPES):
            sq = tuple(d for d in s if d != 1)
            G = structured_grads(sq, 1, seed * 1000 + 19 * t + i)[0].reshape(s) * (grad_scale / 0.3)
            cs.append(G)
            out[f"t{t}_g{i}"] = npy(G)

        def closure():          # linear loss: its gradient is exactly cs, whatever the parameters are
            return sum((p * c).sum() for p, c in zip(params, cs))

        force = [force_gate[t]] if force_gate is not None else None
        with Recorder(force_rand=force) as r:
            opt.step(closure)
        # recorded stream: the step's gate first (psgd.py:615), then -- if the step updates -- per tensor: randn_like(G),
        # per dense factor its subspace draws (spd, and skh for Q0.5EQ1.5), rand([]) of the balancing
        out[f"t{t}_ndraws"] = np.asarray(len(r.draws))
        for k, (kind, x) in enumerate(r.draws):
            out[f"t{t}_draw{k}"] = npy(x)
            out[f"t{t}_draw{k}_kind"] = np.asarray(kind)
        for i, p in enumerate(params):
            out[f"t{t}_p{i}"] = npy(p.data)
            if opt._ms is not None:
                out[f"t{t}_m{i}"] = npy(opt._ms[i])
            for j, (q, ell) in enumerate(zip(*opt._QLs_exprs[i][0])):
                out[f"t{t}_p{i}_Q{j}"] = npy(q)
                out[f"t{t}_p{i}_L{j}"] = npy(ell)
    save("kronwhiten_" + name, out)


def gen_kronwhiten():
    gen_kronwhiten_case("momentum_onthefly", seed=1, momentum=0.9, whiten_grad=False, preconditioner_init_scale=None,
                        lr_params=0.01, lr_preconditioner=0.3)
    gen_kronwhiten_case("nomomentum_last_clip", seed=2, momentum=0.0, whiten_grad=True, preconditioner_init_scale=0.7,
                        update_preconditioner_first=False, grad_scale=6.0, grad_clip_max_amps=(1.2, 2.0), lr_params=0.01,
                        lr_preconditioner=0.5)
    gen_kronwhiten_case("quad4p_momentum_whitengrad", seed=3, dQ="QUAD4P", momentum=0.9, whiten_grad=True,
                        preconditioner_init_scale=None, lr_params=0.01, lr_preconditioner=0.2)
    gen_kronwhiten_case("prob_skewinf", seed=4, T=6, preconditioner_update_probability=0.5, preconditioner_max_skew=float("inf"),
                        preconditioner_init_scale=1.0, momentum=0.5, whiten_grad=False, lr_params=0.01, lr_preconditioner=0.3,
                        force_gate=[0.1, 0.9, 0.2, 0.8, 0.3, 0.7])


# ------------------------------------------------------------------------------------------------
# D. LRA functional + LRAWhiten.step
# ------------------------------------------------------------------------------------------------
def gen_lra_case(name, N, r, dtypes, T=4, lr=0.1, betaL=0.9, damping=1e-9, seed=0, prefix="lra_"):
    out = {"N": np.asarray(N), "r": np.asarray(r), "T": np.asarray(T), "lr": np.asarray(lr),
           "betaL": np.asarray(betaL), "damping": np.asarray(damping)}
    g = torch.Generator().manual_seed(70 + seed)
    U0 = torch.randn(N, r, generator=g)
    U0 = U0 * (0.1 ** 0.5 / max(float(torch.linalg.vector_norm(U0)), 1e-30))
    V0 = torch.randn(N, r, generator=g)
    V0 = V0 * (0.1 ** 0.5 / max(float(torch.linalg.vector_norm(V0)), 1e-30))
    d0 = 0.5 + torch.rand(N, 1, generator=g)
    hscale = 0.5 + 2 * torch.rand(N, 1, generator=g)
    gs = [hscale * torch.randn(N, 1, generator=g) for _ in range(T)]
    out["U0"], out["V0"], out["d0"] = npy(U0), npy(V0), npy(d0)
    for t in range(T):
        out[f"g{t}"] = npy(gs[t])
    for dn in dtypes:
        dt = DT[dn]
        UVd = [U0.to(dt).clone(), V0.to(dt).clone(), d0.to(dt).clone()]
        Luvd = [psgd.lift2single(torch.zeros([], dtype=dt)) for _ in range(3)]
        torch.manual_seed(3000 + seed)
        for t in range(T):
            gt = gs[t].to(dt)
            with Recorder(force_rand=[0.25 if t % 2 == 0 else 0.75]) as r_:
                psgd.update_precond_lra_whiten(UVd, Luvd, gt, lr=lr, betaL=betaL, damping=damping)
            assert len(r_.draws) == 2
            out[f"{dn}_t{t}_vnoise"] = npy(r_.draws[0][1])
            out[f"{dn}_t{t}_coin"] = npy(r_.draws[1][1])
            h = psgd.precond_grad_lra(UVd, gt)
            out[f"{dn}_t{t}_h"] = npy(h)
            out[f"{dn}_t{t}_U"], out[f"{dn}_t{t}_V"], out[f"{dn}_t{t}_d"] = npy(UVd[0]), npy(UVd[1]), npy(UVd[2])
            for k, nm in enumerate(("Lu", "Lv", "Ld")):
                out[f"{dn}_t{t}_{nm}"] = npy(Luvd[k])
    save(prefix + name, out)


def gen_lrawhiten_case(name, T=4, seed=0, shapes=((20, 10), (10,), (3, 4, 5)), dtype=torch.float32, **kw):
    shapes = [tuple(s) for s in shapes]
    out = {"T": np.asarray(T)}
    for k, v in kw.items():
        out["kw_" + k] = np.asarray(v if v is not None else float("nan"))
    g = torch.Generator().manual_seed(800 + seed)
    params = [torch.nn.Parameter((0.5 * torch.randn(*s, generator=g)).to(dtype)) for s in shapes]
    for i, p in enumerate(params):
        out[f"p{i}_init"] = npy(p.data)
    torch.manual_seed(4000 + seed)
    with Recorder() as r0:
        opt = psgd.LRAWhiten(params, **kw)
    out["U0"], out["V0"] = npy(opt._UVd[0]), npy(opt._UVd[1])
    for t in range(T):
        cs = [((0.5 + i) * torch.randn(*s, generator=g)).to(dtype) for i, s in enumerate(shapes)]
        for i, c in enumerate(cs):
            out[f"t{t}_g{i}"] = npy(c)

        def closure():
            return sum((p * c).sum() for p, c in zip(params, cs))

        with Recorder() as r:
            opt.step(closure)
        out[f"t{t}_ndraws"] = np.asarray(len(r.draws))
        for k, (kind, x) in enumerate(r.draws):
            out[f"t{t}_draw{k}"] = npy(x)
            out[f"t{t}_draw{k}_kind"] = np.asarray(kind)
        for i, p in enumerate(params):
            out[f"t{t}_p{i}"] = npy(p.data)
        out[f"t{t}_U"], out[f"t{t}_V"], out[f"t{t}_d"] = npy(opt._UVd[0]), npy(opt._UVd[1]), npy(opt._UVd[2])
    save("lrawhiten_" + name, out)


def gen_lra():
    gen_lra_case("n10_r5", 10, 5, ("fp64", "fp32"), T=6, seed=1)
    gen_lra_case("n2048_r10", 2048, 10, ("fp64", "fp32", "bf16"), T=3, seed=2)
    gen_lra_case("n257_r1", 257, 1, ("fp64", "fp32"), T=4, lr=0.3, betaL=0.5, damping=1e-3, seed=3)
    gen_lra_case("n300_r0", 300, 0, ("fp64", "fp32", "bf16"), T=3, seed=4)          # rank 0 = diagonal preconditioner
    gen_lra_case("n1000_r16", 1000, 16, ("fp32", "bf16"), T=3, seed=5)               # the widest rank of the one-thread-per-row class
    # the wider rank classes of the HIP kernels (two / four threads per row: r <= 32 / <= 64)
    gen_lra_case("n600_r32", 600, 32, ("fp64", "fp32"), T=3, seed=6, prefix="lrabig_")
    gen_lra_case("n1100_r48", 1100, 48, ("fp32", "bf16"), T=2, seed=7, prefix="lrabig_")
    # above the three rank classes: the general path of the HIP kernels (64 < r <= 1024; the reference takes any rank, psgd.py:1113)
    gen_lra_case("n400_r96", 400, 96, ("fp32", "bf16"), T=2, seed=8, prefix="lrabig_")
    gen_lra_case("n300_r130", 300, 130, ("fp64", "fp32"), T=2, seed=9, prefix="lrabig_")
    gen_lrawhiten_case("grad_r5", seed=1, rank_of_approximation=5, preconditioner_init_scale=1.0)
    gen_lrawhiten_case("momentum_r3_last", seed=2, rank_of_approximation=3, preconditioner_init_scale=None,
                       momentum=0.9, whiten_grad=False, update_preconditioner_first=False, lr_params=0.01)
    # RMS clip engaged (rms(h) ~ 9 x rms(g) > 2) on steps whose update runs AFTER the apply (psgd.py:1172-1183)
    gen_lrawhiten_case("clip_last_r4", seed=4, rank_of_approximation=4, preconditioner_init_scale=3.0,
                       update_preconditioner_first=False, lr_params=0.01)
    gen_lrawhiten_case("momentum_r32", seed=3, rank_of_approximation=32, preconditioner_init_scale=1.0, momentum=0.9)
    gen_lrawhiten_bf16()


def gen_lrawhiten_bf16():
    # bf16 parameters (the reference keeps U, V, d in the parameter dtype: psgd.py:1113-1128), N = 1710 >= 512, even rank: the HIP engine's packed
    # two-rows-per-thread passes (csrc/kernels_lra_pk.hiph) over three whole blocks + a 174-row tail; momentum, both update orders
    gen_lrawhiten_case("bf16_r10_n1710", T=4, seed=11, shapes=((40, 30), (30,), (6, 8, 10)), dtype=torch.bfloat16,
                       rank_of_approximation=10, preconditioner_init_scale=1.0, momentum=0.9, lr_params=0.01)
    gen_lrawhiten_case("bf16_r4_n1710_last", T=4, seed=12, shapes=((40, 30), (30,), (6, 8, 10)), dtype=torch.bfloat16,
                       rank_of_approximation=4, preconditioner_init_scale=None, update_preconditioner_first=False, lr_params=0.01)


if __name__ == "__main__":
    torch.set_num_threads(4)
    groups = {"helpers": gen_helpers, "kron": gen_kron, "kron_eq": gen_kron_eq, "kron_geoms": gen_kron_geoms,
              "kron_pro4p": gen_kron_pro4p, "kwns4": gen_kwns4, "kronwhiten": gen_kronwhiten, "lra": gen_lra,
              "lrawhiten_bf16": gen_lrawhiten_bf16}
    for name in (sys.argv[1:] or list(groups)):       # no argument: everything
        groups[name]()
