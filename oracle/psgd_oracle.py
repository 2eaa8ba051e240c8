"""
CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A plain torch-CPU restatement of the one hot path of lixilinx/psgd_torch that this repository
accelerates (PSGD Kron "Q0.5EQ1.5" whitening update + apply, the LRA update + apply, and the two
host shells KWNS4.step / LRAWhiten.step that call them).  It exists to CHECK the HIP engine:

  * only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import it;
  * nothing under psgd_torch_amd/ imports it, and the product path never falls back to it.

Pinning: every function below is validated against golden vectors captured from the reference
itself (tests/golden/*.npz, produced by tests/golden/gen_golden.py which imports /root/reference
with an `opt_einsum` stand-in -- the reference's only third-party arithmetic dependency, which is
absent from this image and un-pinned by the reference; it contributes contraction ORDER only).
See tests/test_oracle_golden.py.

Differences from the reference, on purpose:
  * no einsum dependency: contractions are explicit mode products (torch.matmul on reshaped views);
  * every random draw is an explicit argument (`noise`), so a run is a pure function of its inputs;
  * real dtypes only (fp64 / fp32 / bf16), like the reference's KWNS4 wrapper
    (wrapped_as_torch_optimizer_for_ddp.py:7).

Each function cites the reference lines it restates (file:line under /root/reference).
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------------
def lift2single(x: Tensor) -> Tensor:
    """psgd.py:96-98 -- lift half precisions to fp32, keep fp32/fp64."""
    return x.to(torch.float32) if torch.finfo(x.dtype).eps > 1e-6 else x


def kron_factor_kinds(shape: Sequence[int], max_size: float = float("inf"), max_skew: float = 1.0) -> List[str]:
    """psgd.py:208 -- per-dim dense/diag decision of init_kron.  A 0-dim tensor has one 'scalar' factor
    (psgd.py:189-195)."""
    if len(shape) == 0:
        return ["scalar"]
    numel = 1
    for s in shape:
        numel *= s
    kinds = []
    for size in shape:
        if size <= 1 or size > max_size or size ** 2 > max_skew * numel:
            kinds.append("diag")
        else:
            kinds.append("dense")
    return kinds


def init_kron(t: Tensor, Scale: float = 1.0, max_size: float = float("inf"), max_skew: float = 1.0):
    """psgd.py:161-263 (dQ='Q0.5EQ1.5' branch): returns [Q list, L list], kinds.
    Q_i = Scale^(1/k) * I (dense) or * ones (diag); L_i = fp32/fp64 0-dim zeros (psgd.py:207)."""
    shape = tuple(t.shape)
    if len(shape) == 0:
        Q = [Scale * torch.ones_like(t)]
        L = [lift2single(torch.zeros_like(t))]
        return [Q, L], ["scalar"]
    if len(shape) > 26:
        raise ValueError(f"Got tensor with dim {len(shape)}; einsum runs out of letters; replace 26 with larger numbers.")
    scale = Scale ** (1 / len(shape))
    kinds = kron_factor_kinds(shape, max_size, max_skew)
    Q, L = [], []
    for size, kind in zip(shape, kinds):
        L.append(lift2single(torch.zeros([], dtype=t.dtype)))
        if kind == "diag":
            Q.append(scale * torch.ones(size, dtype=t.dtype))
        else:
            Q.append(scale * torch.eye(size, dtype=t.dtype))
    return [Q, L], kinds


def _mode_product(M: Tensor, X: Tensor, i: int) -> Tensor:
    """Y[..., a, ...] = sum_b M[a, b] X[..., b, ...] along dim i (dense M) -- one factor of exprP/exprA."""
    Xm = X.movedim(i, 0)
    shp = Xm.shape
    Y = M @ Xm.reshape(shp[0], -1)
    return Y.reshape((M.shape[0],) + tuple(shp[1:])).movedim(0, i)


def _mode_scale(d: Tensor, X: Tensor, i: int) -> Tensor:
    """Y[..., a, ...] = d[a] X[..., a, ...] along dim i (diagonal factor)."""
    view = [1] * X.dim()
    view[i] = -1
    return X * d.reshape(view)


def precond_grad_kron(Q: List[Tensor], G: Tensor) -> Tensor:
    """psgd.py:322-327 + exprP (psgd.py:251-252): h = (kron_i Q_i^T Q_i) G, applied as Q then Q^T per mode
    (the defining order; any pairwise order is equally 'the reference' -- SURVEY 8c)."""
    if G.dim() == 0:
        return Q[0] * Q[0] * G
    X = G
    for i, q in enumerate(Q):
        if q.dim() < 2:
            X = _mode_scale(q * q, X, i)
        else:
            X = _mode_product(q.t(), _mode_product(q, X, i), i)
    return X


def gram_mode(Pg: Tensor, i: int, dense: bool) -> Tensor:
    """exprGs[i] (psgd.py:221-223 diag, 240-243 dense): contraction of Pg with itself keeping dim i."""
    if Pg.dim() == 0:
        return Pg * Pg
    Xm = Pg.movedim(i, 0).reshape(Pg.shape[i], -1)
    if dense:
        return Xm @ Xm.t()
    return (Xm * Xm).sum(dim=1)


# --------------------------------------------------------------------------------------------
# spectral-norm lower bounds, Procrustes step
# --------------------------------------------------------------------------------------------
def _subspace_iteration(A: Tensor, normalizing_factor: Tensor, noise: Tensor, half_iters: int = 2) -> Tensor:
    """Shared body of psgd.py:60-68 / 85-93."""
    smallest_normal = torch.finfo(A.dtype).smallest_normal
    A = A / normalizing_factor
    j = torch.argmax(torch.linalg.vector_norm(A, dim=1))
    V = noise.to(A.dtype)
    V = A[j] + torch.sgn(torch.sum(A[j] * V, dim=1, keepdim=True)) * V
    for _ in range(half_iters):
        V = V @ A
        V = V / (torch.linalg.vector_norm(V, dim=1, keepdim=True) + smallest_normal)
        V = V @ A
    return normalizing_factor * torch.amax(torch.linalg.vector_norm(V, dim=1))


def norm_lower_bound_spd(A: Tensor, noise: Tensor, half_iters: int = 2) -> Tensor:
    """psgd.py:46-68.  `noise` is the (k, n) Gaussian draw of psgd.py:62 (k=32 at every call site)."""
    smallest_normal = torch.finfo(A.dtype).smallest_normal
    normalizing_factor = A.diagonal().amax() + smallest_normal
    return _subspace_iteration(A, normalizing_factor, noise, half_iters)


def norm_lower_bound_skh(A: Tensor, noise: Tensor, half_iters: int = 2) -> Tensor:
    """psgd.py:71-93.  `noise` is the (k, n) Gaussian draw of psgd.py:87."""
    smallest_normal = torch.finfo(A.dtype).smallest_normal
    normalizing_factor = A.abs().amax() + smallest_normal
    return _subspace_iteration(A, normalizing_factor, noise, half_iters)


def procrustes_step2(Q: Tensor, noise: Tensor, max_step_size: float = 1 / 8) -> None:
    """psgd.py:101-124, in place on Q."""
    R = Q.t() - Q
    R = R / (norm_lower_bound_skh(R, noise) + torch.finfo(R.dtype).smallest_normal)
    RQ = R @ Q
    RRQ = R @ RQ
    tr_RQ = RQ.diagonal().sum()
    tr_RRQ = RRQ.diagonal().sum()
    a = torch.where(tr_RRQ < 0, torch.clamp(-tr_RQ / tr_RRQ, max=max_step_size), max_step_size)
    Q.add_(a * (RQ + 0.5 * a * RRQ))


def procrustes_step3(Q: Tensor, noise: Tensor, max_step_size: float = 1 / 3) -> None:
    """psgd.py:127-158, in place on Q (third-order expansion of the rotation; used by the PRO4P geometry)."""
    R = Q.t() - Q
    R = R / (norm_lower_bound_skh(R, noise) + torch.finfo(R.dtype).smallest_normal)
    RQ = R @ Q
    RRQ = R @ RQ
    RRRQ = R @ RRQ
    tr_RQ = RQ.diagonal().sum()
    tr_RRQ = RRQ.diagonal().sum()
    tr_RRRQ = RRRQ.diagonal().sum()
    if tr_RQ > 0 and tr_RRRQ < 0:
        if torch.finfo(tr_RQ.dtype).eps > 1e-6:
            tr_RQ, tr_RRQ, tr_RRRQ = tr_RQ.to(torch.float32), tr_RRQ.to(torch.float32), tr_RRRQ.to(torch.float32)
        a = (-tr_RRQ - torch.sqrt(tr_RRQ * tr_RRQ - 1.5 * tr_RQ * tr_RRRQ)) / (0.75 * tr_RRRQ)
        a = torch.clamp(a, max=max_step_size)
        Q.add_(a * (RQ + 0.5 * a * (RRQ + 0.25 * a * RRRQ)))


def balance_kron_precond(Q: List[Tensor]) -> None:
    """psgd.py:266-275."""
    order = len(Q)
    if order > 1:
        norms = [torch.max(torch.abs(q)) for q in Q]
        gmean = torch.prod(torch.stack(norms)) ** (1 / order)
        for i, q in enumerate(Q):
            q.mul_(gmean / norms[i])


# --------------------------------------------------------------------------------------------
# Kron whitening update, dQ = Q^0.5 E Q^1.5
# --------------------------------------------------------------------------------------------
class KronNoise:
    """The random draws of ONE call of update_precond_kron_whiten_q0p5eq1p5, in reference draw order
    (SURVEY 8c): randn_like(G) psgd.py:403; per DENSE factor randn(32,d) psgd.py:62 then randn(32,d)
    psgd.py:87; finally rand([]) psgd.py:418 (balance gate)."""

    def __init__(self, g_noise: Tensor, spd: List[Optional[Tensor]], skh: List[Optional[Tensor]], balance_u: float):
        self.g_noise, self.spd, self.skh, self.balance_u = g_noise, spd, skh, balance_u

    @staticmethod
    def draw(G: Tensor, kinds: Sequence[str], gen: torch.Generator, k: int = 32) -> "KronNoise":
        g_noise = torch.randn(G.shape, generator=gen, dtype=torch.float32).to(G.dtype)
        spd, skh = [], []
        for kind, size in zip(kinds, G.shape if G.dim() else [1]):
            if kind == "dense":
                spd.append(torch.randn(k, size, generator=gen, dtype=torch.float32).to(G.dtype))
                skh.append(torch.randn(k, size, generator=gen, dtype=torch.float32).to(G.dtype))
            else:
                spd.append(None)
                skh.append(None)
        u = float(torch.rand([], generator=gen))
        return KronNoise(g_noise, spd, skh, u)


def update_precond_kron_whiten_q0p5eq1p5(QL, G: Tensor, noise: KronNoise, lr: float = 0.1, betaL: float = 0.9,
                                         damping: float = 1e-9, return_intermediates: bool = False):
    """psgd.py:394-419, in place on Q and L."""
    Q, L = QL
    total_numel = G.numel()
    damp = damping + torch.finfo(G.dtype).eps * G.abs()
    Pg = precond_grad_kron(Q, G + damp * noise.g_noise.to(G.dtype))
    inter = {"Pg": Pg.clone(), "term1": []} if return_intermediates else None
    for i, q in enumerate(Q):
        dense = q.dim() >= 2
        term1 = gram_mode(Pg, i, dense)
        if inter is not None:
            inter["term1"].append(term1.clone())
        if not dense:
            term2 = total_numel / q.numel()
            ell = torch.max(term1) + term2
            L[i].copy_(torch.max(betaL * L[i] + (1 - betaL) * ell, ell))
            q.mul_(1 - lr / L[i] * (term1 - term2))
        else:
            term2 = total_numel / q.shape[0]
            ell = norm_lower_bound_spd(term1, noise.spd[i]) + term2
            L[i].copy_(torch.max(betaL * L[i] + (1 - betaL) * ell, ell))
            q.sub_(lr / L[i] * (term1 @ q - term2 * q))
            procrustes_step2(q, noise.skh[i])
    if noise.balance_u < 0.01:
        balance_kron_precond(Q)
    return inter


def _whiten_terms(QL, G: Tensor, noise: KronNoise, damping: float):
    """Shared head of psgd.py:367-392 / 455-483: Pg = (kron Q^T Q)(G + damped noise)."""
    damp = damping + torch.finfo(G.dtype).eps * G.abs()
    return precond_grad_kron(QL[0], G + damp * noise.g_noise.to(G.dtype))


def update_precond_kron_whiten_qeq(QL, G: Tensor, noise: KronNoise, lr: float = 0.1, betaL: float = 0.9,
                                   damping: float = 1e-9) -> None:
    """psgd.py:367-391 (dQ = Q*E*Q), in place on Q and L; noise.skh is unused."""
    Q, L = QL
    total_numel = G.numel()
    Pg = _whiten_terms(QL, G, noise, damping)
    for i, q in enumerate(Q):
        dense = q.dim() >= 2
        term1 = gram_mode(Pg, i, dense)
        if not dense:
            term2 = total_numel / q.numel()
            ell = torch.max(term1) + term2
            L[i].copy_(torch.max(betaL * L[i] + (1 - betaL) * ell, ell))
            q.mul_(1 - lr / L[i] * (term1 - term2))
        else:
            term2 = total_numel / q.shape[0]
            ell = norm_lower_bound_spd(term1, noise.spd[i]) + term2
            L[i].copy_(torch.max(betaL * L[i] + (1 - betaL) * ell, ell))
            q.sub_(lr / L[i] * (q @ term1 - q * term2))
    if noise.balance_u < 0.01:
        balance_kron_precond(Q)


def update_precond_kron_whiten_quad(QL, G: Tensor, noise: KronNoise, lr: float = 0.1, betaL: float = 0.9,
                                    damping: float = 1e-9) -> None:
    """psgd.py:455-483 (quadratic form; Q stays symmetric), in place on Q and L; noise.skh is unused."""
    Q, L = QL
    total_numel = G.numel()
    Pg = _whiten_terms(QL, G, noise, damping)
    for i, q in enumerate(Q):
        dense = q.dim() >= 2
        term1 = gram_mode(Pg, i, dense)
        if not dense:
            term2 = total_numel / q.numel()
            ell = torch.max(term1) + term2
            L[i].copy_(torch.max(betaL * L[i] + (1 - betaL) * ell, ell))
            gain = 1 - lr / 2 / L[i] * (term1 - term2)
            q.mul_(gain * gain)
        else:
            term2 = total_numel / q.shape[0]
            ell = norm_lower_bound_spd(term1, noise.spd[i]) + term2
            L[i].copy_(torch.max(betaL * L[i] + (1 - betaL) * ell, ell))
            p = q - lr / 2 / L[i] * (term1 @ q - term2 * q)
            p = p - lr / 2 / L[i] * (p @ term1 - p * term2)
            q.copy_((p + p.t()) / 2)
    if noise.balance_u < 0.01:
        balance_kron_precond(Q)


def update_precond_kron_whiten_qep(QL, G: Tensor, noise: KronNoise, lr: float = 0.1, betaL: float = 0.9,
                                   damping: float = 1e-9) -> None:
    """psgd.py:339-364 (dQ = Q*E*P), in place on Q and L.  Balancing runs on every call, BEFORE the update
    (psgd.py:346-347); noise.skh and noise.balance_u are unused."""
    Q, L = QL
    balance_kron_precond(Q)
    total_numel = G.numel()
    Pg = _whiten_terms(QL, G, noise, damping)
    for i, q in enumerate(Q):
        dense = q.dim() >= 2
        QPg = (Pg * q if Pg.dim() == 0 else _mode_scale(q, Pg, i)) if not dense else _mode_product(q, Pg, i)   # exprQs[i]
        term1 = gram_mode(QPg, i, dense)
        if not dense:
            term2 = total_numel / q.numel() * q * q
            ell = torch.max(term1 + term2)
            L[i].copy_(torch.max(betaL * L[i] + (1 - betaL) * ell, ell))
            q.mul_(1 - lr / L[i] * (term1 - term2))
        else:
            term2 = total_numel / q.shape[0] * q @ q.t()
            ell = norm_lower_bound_spd(term1 + term2, noise.spd[i])
            L[i].copy_(torch.max(betaL * L[i] + (1 - betaL) * ell, ell))
            q.sub_(lr / L[i] * (term1 - term2) @ q)


def update_precond_kron_whiten_quad4p(QL, G: Tensor, noise: KronNoise, lr: float = 0.1, betaL: float = 0.9,
                                      damping: float = 1e-9) -> None:
    """psgd.py:486-513: as QUAD but the factors are P itself (applied once, exprA) and the steps are full lr/L."""
    Q, L = QL
    total_numel = G.numel()
    damp = damping + torch.finfo(G.dtype).eps * G.abs()
    Pg = apply_q_kron(Q, G + damp * noise.g_noise.to(G.dtype))
    for i, q in enumerate(Q):
        dense = q.dim() >= 2
        term1 = gram_mode(Pg, i, dense)
        if not dense:
            term2 = total_numel / q.numel()
            ell = torch.max(term1) + term2
            L[i].copy_(torch.max(betaL * L[i] + (1 - betaL) * ell, ell))
            gain = 1 - lr / L[i] * (term1 - term2)
            q.mul_(gain * gain)
        else:
            term2 = total_numel / q.shape[0]
            ell = norm_lower_bound_spd(term1, noise.spd[i]) + term2
            L[i].copy_(torch.max(betaL * L[i] + (1 - betaL) * ell, ell))
            p = q - lr / L[i] * (term1 @ q - term2 * q)
            p = p - lr / L[i] * (p @ term1 - p * term2)
            q.copy_((p + p.t()) / 2)
    if noise.balance_u < 0.01:
        balance_kron_precond(Q)


def update_precond_kron_whiten_pro4p(QL, G: Tensor, noise: KronNoise, pro_noise, lr: float = 0.1, betaL: float = 0.9,
                                     damping: float = 1e-9) -> List[int]:
    """psgd.py:422-452: fits P directly with dP = P^0.5 E P; after the gradient step up to 10 procrustes_step3 rotations
    bring each dense factor back to (almost) Hermitian.  pro_noise[i] = list of the (32, d) draws of the successive
    procrustes_step3 calls of dense factor i (at least as many as get used).  Returns the number of rotations per factor."""
    Q, L = QL
    total_numel = G.numel()
    damp = damping + torch.finfo(G.dtype).eps * G.abs()
    Pg = apply_q_kron(Q, G + damp * noise.g_noise.to(G.dtype))
    used = []
    for i, q in enumerate(Q):
        dense = q.dim() >= 2
        term1 = gram_mode(Pg, i, dense)
        if not dense:
            term2 = total_numel / q.numel()
            ell = torch.max(term1) + term2
            L[i].copy_(torch.max(betaL * L[i] + (1 - betaL) * ell, ell))
            q.mul_(1 - lr / L[i] * (term1 - term2))
            used.append(0)
        else:
            term2 = total_numel / q.shape[0]
            ell = norm_lower_bound_spd(term1, noise.spd[i]) + term2
            L[i].copy_(torch.max(betaL * L[i] + (1 - betaL) * ell, ell))
            q.sub_(lr / L[i] * (term1 @ q - term2 * q))
            n = 0
            for k in range(10):
                procrustes_step3(q, pro_noise[i][k].to(q.dtype))
                n += 1
                if (q.t() - q).abs().amax() < 0.001 * q.abs().amax():
                    break
            used.append(n)
    if noise.balance_u < 0.01:
        balance_kron_precond(Q)
    return used


def precond_grad_kron_4p(Q: List[Tensor], G: Tensor) -> Tensor:
    """psgd.py:573: for the geometries that fit P directly the preconditioned gradient is exprA(*Q, G)."""
    return apply_q_kron(Q, G)


def apply_q_kron(Q: List[Tensor], X: Tensor) -> Tensor:
    """exprA (psgd.py:248-249): A = (kron_i Q_i) X, one factor per mode."""
    if X.dim() == 0:
        return Q[0] * X
    for i, q in enumerate(Q):
        X = _mode_scale(q, X, i) if q.dim() < 2 else _mode_product(q, X, i)
    return X


def solve_q_kron(Q: List[Tensor], V: Tensor) -> Tensor:
    """psgd.py:288-303: conjB = V x_i Q_i^{-T} -- per mode a right triangular solve (in fp32 for bf16, lift2single,
    rounded back to V's dtype after every mode) or a division by the diagonal factor.  Returned in V's own dim order
    (the reference carries it cyclically permuted; its Grams are order independent)."""
    if V.dim() == 0:
        return V / Q[0]
    B = V
    for i, q in enumerate(Q):
        Bm = B.movedim(i, -1)
        if q.dim() < 2:
            Bm = Bm / q
        else:
            shp = Bm.shape
            Y = torch.linalg.solve_triangular(lift2single(q), lift2single(Bm.reshape(-1, shp[-1])), upper=True, left=False)
            Bm = Y.to(B.dtype).reshape(shp)
        B = Bm.movedim(-1, i)
    return B


def update_precond_kron_eq(QL, V: Tensor, Hvp: Tensor, spd_noise: List[Optional[Tensor]], balance_u: float,
                           lr: float = 0.1, betaL: float = 0.9) -> None:
    """psgd.py:278-319, in place on Q and L: the triangular geometry dQ = E*Q."""
    Q, L = QL
    A = apply_q_kron(Q, Hvp)
    B = solve_q_kron(Q, V)
    for i, q in enumerate(Q):
        dense = q.dim() >= 2
        term1 = gram_mode(A, i, dense)
        term2 = gram_mode(B, i, dense)
        if not dense:
            ell = torch.max(term1 + term2)
            L[i].copy_(torch.max(betaL * L[i] + (1 - betaL) * ell, ell))
            q.sub_(lr / L[i] * (term1 - term2) * q)
        else:
            ell = norm_lower_bound_spd(term1 + term2, spd_noise[i])
            L[i].copy_(torch.max(betaL * L[i] + (1 - betaL) * ell, ell))
            q.sub_(lr / L[i] * torch.triu(term1 - term2) @ q)
    if balance_u < 0.01:
        balance_kron_precond(Q)


def update_precond_kron_whiten_eq(QL, G: Tensor, noise: KronNoise, lr: float = 0.1, betaL: float = 0.9,
                                  damping: float = 1e-9) -> None:
    """psgd.py:330-336: V = noise.g_noise is both the probe and the damping noise; noise.skh is unused."""
    V = noise.g_noise.to(G.dtype)
    damp = damping + torch.finfo(G.dtype).eps * G.abs()
    update_precond_kron_eq(QL, V, G + damp * V, noise.spd, noise.balance_u, lr=lr, betaL=betaL)


# --------------------------------------------------------------------------------------------
# LRA preconditioner
# --------------------------------------------------------------------------------------------
def IpUVtmatvec(U: Tensor, V: Tensor, x: Tensor) -> Tensor:
    """psgd.py:987-991."""
    return x + U.mm(V.t().mm(x))


def update_precond_lra(UVd, Luvd, v: Tensor, h: Tensor, coin_u: float, lr: float = 0.1, betaL: float = 0.9) -> None:
    """psgd.py:994-1052, in place.  `coin_u` is the rand([]) draw of psgd.py:1035."""
    U, V, d = UVd
    Lu, Lv, Ld = Luvd
    UtU, VtV = U.t() @ U, V.t() @ V
    trUtU, trVtV = torch.sum(UtU.diagonal()), torch.sum(VtV.diagonal())
    rho = (trUtU / trVtV) ** (1 / 4)
    rho2 = rho * rho
    E = 0.1 * (UtU / rho2 - VtV * rho2) / (trUtU / rho2 + trVtV * rho2)
    E2 = 0.5 * E @ E
    U.div_(rho)
    V.mul_(rho)
    U.sub_(U @ (E - E2))
    V.add_(V @ (E + E2))

    Qh = IpUVtmatvec(U, V, d * h)
    Ph = d * IpUVtmatvec(V, U, Qh)

    IpVtU = V.t().mm(U)
    IpVtU.diagonal().add_(1)
    invQtv = v / d
    LU, pivots = torch.linalg.lu_factor(lift2single(IpVtU))
    invQtv = invQtv - V.mm(torch.linalg.lu_solve(LU, pivots, lift2single(U.t().mm(invQtv)), adjoint=True).to(V.dtype))
    invPv = invQtv - U.mm(torch.linalg.lu_solve(LU, pivots, lift2single(V.t().mm(invQtv))).to(U.dtype))
    invPv = invPv / d

    Phh, vinvPv = Ph * h, v * invPv
    ell = torch.max(torch.abs(Phh)) + torch.max(torch.abs(vinvPv))
    Ld.copy_(torch.max(betaL * Ld + (1 - betaL) * ell, ell))
    d.sub_(lr / Ld * (Phh - vinvPv) * d)

    a, b = Qh, invQtv
    if coin_u < 0.5:
        atV = a.t().mm(V)
        btV = b.t().mm(V)
        atVVt = atV.mm(V.t())
        btVVt = btV.mm(V.t())
        ell = (torch.linalg.vector_norm(a) * torch.linalg.vector_norm(atVVt)
               + torch.linalg.vector_norm(b) * torch.linalg.vector_norm(btVVt))
        Lu.copy_(torch.max(betaL * Lu + (1 - betaL) * ell, ell))
        U.sub_(lr / Lu * (a.mm(atV.mm(IpVtU)) - b.mm(btV.mm(IpVtU))))
    else:
        atU = a.t().mm(U)
        btU = b.t().mm(U)
        UUta = U.mm(atU.t())
        UUtb = U.mm(btU.t())
        ell = (torch.linalg.vector_norm(a) * torch.linalg.vector_norm(UUta)
               + torch.linalg.vector_norm(b) * torch.linalg.vector_norm(UUtb))
        Lv.copy_(torch.max(betaL * Lv + (1 - betaL) * ell, ell))
        V.sub_(lr / Lv * ((a + V.mm(atU.t())).mm(atU) - (b + V.mm(btU.t())).mm(btU)))


def precond_grad_lra(UVd, g: Tensor) -> Tensor:
    """psgd.py:1055-1063."""
    U, V, d = UVd
    g = IpUVtmatvec(U, V, d * g)
    g = d * IpUVtmatvec(V, U, g)
    return g


def update_precond_lra_whiten(UVd, Luvd, g: Tensor, v_noise: Tensor, coin_u: float, lr: float = 0.1,
                              betaL: float = 0.9, damping: float = 1e-9) -> None:
    """psgd.py:1066-1072.  `v_noise` is the randn_like(g) draw of psgd.py:1070."""
    v = v_noise.to(g.dtype)
    damp = damping + torch.finfo(g.dtype).eps * g.abs()
    update_precond_lra(UVd, Luvd, v, g + damp * v, coin_u, lr=lr, betaL=betaL)


# --------------------------------------------------------------------------------------------
# host shells restated (callers of the path): KWNS4.step and LRAWhiten.step
# --------------------------------------------------------------------------------------------
class KWNS4Oracle:
    """wrapped_as_torch_optimizer_for_ddp.py:25-176 restated for ONE param group on CPU tensors.

    `uniform()` supplies the host Bernoulli draws (rand([]) at ..._ddp.py:110) and `noise_for(G, kinds)`
    the per-update KronNoise, so that a test can replay the reference's recorded draws.
    """

    def __init__(self, params: List[Tensor], whiten_grad=False, preconditioner_max_size=float("inf"),
                 preconditioner_max_skew=1.0, preconditioner_init_scale=1.0, lr_params=2e-4, lr_preconditioner=0.5,
                 betaL=0.9, damping=1e-9, momentum=0.9, weight_decay=0.05, decoupled_weight_decay=True,
                 grad_clip_max_amps=(2.0, 10.0), preconditioner_update_probability=1.0,
                 preconditioner_dtype: Optional[torch.dtype] = torch.bfloat16, update_preconditioner_first=True,
                 uniform: Optional[Callable[[], float]] = None,
                 noise_for: Optional[Callable[[Tensor, Sequence[str]], KronNoise]] = None, seed: int = 0,
                 dQ: str = "Q0.5EQ1.5"):
        # dQ: the three lines ..._ddp.py:84-86 switched the way KronWhiten switches them (psgd.py:565-586): the update function,
        # and for the geometries that fit P itself the apply function and the squared initial scale (psgd.py:186-187)
        self._update = {"Q0.5EQ1.5": update_precond_kron_whiten_q0p5eq1p5, "Q0p5EQ1p5": update_precond_kron_whiten_q0p5eq1p5,
                        "EQ": update_precond_kron_whiten_eq, "QEQ": update_precond_kron_whiten_qeq,
                        "QUAD": update_precond_kron_whiten_quad, "QEP": update_precond_kron_whiten_qep,
                        "QUAD4P": update_precond_kron_whiten_quad4p}[dQ]
        self._p4 = dQ == "QUAD4P"
        self._apply = precond_grad_kron_4p if self._p4 else precond_grad_kron
        self.params = params
        self.g = dict(whiten_grad=whiten_grad, preconditioner_max_size=preconditioner_max_size,
                      preconditioner_max_skew=preconditioner_max_skew,
                      preconditioner_init_scale=preconditioner_init_scale, lr_params=lr_params,
                      lr_preconditioner=lr_preconditioner, betaL=betaL, damping=damping, momentum=momentum,
                      weight_decay=weight_decay, decoupled_weight_decay=decoupled_weight_decay,
                      grad_clip_max_amps=grad_clip_max_amps,
                      preconditioner_update_probability=preconditioner_update_probability,
                      preconditioner_dtype=preconditioner_dtype,
                      update_preconditioner_first=update_preconditioner_first)
        self.state = [dict() for _ in params]
        gen = torch.Generator().manual_seed(seed)
        self._uniform = uniform if uniform is not None else (lambda: float(torch.rand([], generator=gen)))
        self._noise_for = noise_for if noise_for is not None else (lambda G, kinds: KronNoise.draw(G, kinds, gen))

    @torch.no_grad()
    def step(self, grads: List[Optional[Tensor]]) -> None:
        g = self.g
        momentum = g["momentum"]
        max_avg_amp, max_element_amp = g["grad_clip_max_amps"]
        if self._uniform() < g["preconditioner_update_probability"]:      # ..._ddp.py:109-110
            first, last = g["update_preconditioner_first"], not g["update_preconditioner_first"]
        else:
            first, last = False, False
        for p, grad, state in zip(self.params, grads, self.state):
            if grad is None:
                continue
            wd = g["weight_decay"]                                          # ..._ddp.py:117-122
            if wd > 0.0:
                if g["decoupled_weight_decay"]:
                    p.mul_(1.0 - wd * g["lr_params"])
                else:
                    grad = grad.add(p, alpha=wd)
            grad = grad.squeeze()                                           # ..._ddp.py:124-127
            if g["preconditioner_dtype"]:
                grad = grad.to(g["preconditioner_dtype"])
            if len(state) == 0:                                             # ..._ddp.py:129-137
                state["QL"], state["kinds"] = init_kron(grad, Scale=g["preconditioner_init_scale"] ** (2 if self._p4 else 1),
                                                        max_size=g["preconditioner_max_size"],
                                                        max_skew=g["preconditioner_max_skew"])
                state["step"] = 0
                state["ema"] = None if momentum == 0.0 else torch.zeros_like(grad)
            t = state["step"]                                               # ..._ddp.py:139-143
            if momentum > 0.0:
                beta = min(t / (t + 1), momentum)
                state["ema"].mul_(beta).add_(grad, alpha=1.0 - beta)
            state["step"] += 1
            to_be_whitened = grad if g["whiten_grad"] else state["ema"]    # ..._ddp.py:145-148
            if first:
                self._update(state["QL"], to_be_whitened, self._noise_for(to_be_whitened, state["kinds"]),
                             lr=g["lr_preconditioner"], betaL=g["betaL"], damping=g["damping"])
            to_be_preconded = grad if momentum == 0.0 else state["ema"]    # ..._ddp.py:150-151
            h = self._apply(state["QL"][0], to_be_preconded)
            avg_amp = torch.sqrt(torch.mean(h * h))                         # ..._ddp.py:153-157
            if avg_amp > max_avg_amp:
                h = h * (max_avg_amp / avg_amp)
            h = h.clamp(min=-max_element_amp, max=max_element_amp)
            p.subtract_(h.view_as(p).to(p.dtype), alpha=g["lr_params"])
            if last:                                                        # ..._ddp.py:159-161
                self._update(state["QL"], to_be_whitened, self._noise_for(to_be_whitened, state["kinds"]),
                             lr=g["lr_preconditioner"], betaL=g["betaL"], damping=g["damping"])


class KronWhitenOracle:
    """psgd.py:516-654 (KronWhiten) restated with explicit gradients instead of a closure, for dQ in {"Q0.5EQ1.5", "QUAD4P"}.
    `uniform()` supplies the per-step gate (psgd.py:615) and `noise_for(G, kinds)` the KronNoise of each update call, in the
    reference's order: ALL tensors are updated, then ALL are preconditioned (psgd.py:620-639), then clipped and applied per
    tensor (psgd.py:642-651).  The attributes the reference lets users anneal are plain members here too."""

    def __init__(self, params: List[Tensor], preconditioner_max_size=float("inf"), preconditioner_max_skew=1.0,
                 preconditioner_init_scale: Optional[float] = None, lr_params=0.001, lr_preconditioner=0.1, betaL=0.9,
                 damping=1e-9, momentum=0.0, grad_clip_max_amps=(2.0, 10.0), preconditioner_update_probability=1.0,
                 update_preconditioner_first=True, whiten_grad=True, dQ="Q0.5EQ1.5",
                 uniform: Optional[Callable[[], float]] = None,
                 noise_for: Optional[Callable[[Tensor, Sequence[str]], KronNoise]] = None, seed: int = 0):
        assert dQ in ("Q0.5EQ1.5", "Q0p5EQ1p5", "QUAD4P")
        self.params = params
        self.lr_params, self.lr_preconditioner, self.betaL, self.damping = lr_params, lr_preconditioner, betaL, damping
        self.momentum = momentum if (0 < momentum < 1) else 0.0                                       # psgd.py:545
        self.grad_clip_max_amps = grad_clip_max_amps
        self.preconditioner_update_probability = preconditioner_update_probability
        self.update_preconditioner_first = update_preconditioner_first
        self._max_size, self._max_skew, self._dQ = preconditioner_max_size, preconditioner_max_skew, dQ
        self._p4 = dQ == "QUAD4P"
        self.QLs, self.kinds = None, None
        if preconditioner_init_scale is not None:                                                     # psgd.py:558
            self._init([p.squeeze() for p in params], preconditioner_init_scale)
        self.ms, self._counter_m = None, 0
        self._whiten_grad = whiten_grad
        if not whiten_grad:
            assert self.momentum > 0
        gen = torch.Generator().manual_seed(seed)
        self._uniform = uniform if uniform is not None else (lambda: float(torch.rand([], generator=gen)))
        self._noise_for = noise_for if noise_for is not None else (lambda G, kinds: KronNoise.draw(G, kinds, gen))

    def _init(self, like: List[Tensor], scale):
        s = scale ** 2 if self._p4 else scale                                                         # psgd.py:186-187
        both = [init_kron(t, Scale=s, max_size=self._max_size, max_skew=self._max_skew) for t in like]
        self.QLs, self.kinds = [b[0] for b in both], [b[1] for b in both]

    @torch.no_grad()
    def step(self, grads: List[Tensor]) -> None:
        grads = [g.squeeze() for g in grads]                                                          # psgd.py:597
        if self.QLs is None:                                                                          # psgd.py:599-602
            scale = max([torch.mean((torch.abs(g)) ** 4) for g in grads])
            scale = (scale + self.damping ** 4) ** (-1 / 8)
            self._init(grads, scale)
        if self.momentum > 0:                                                                         # psgd.py:604-613
            beta = min(self._counter_m / (1 + self._counter_m), self.momentum)
            self._counter_m += 1
            if self.ms is None:
                self.ms = [torch.zeros_like(g) for g in grads]
            for m, g in zip(self.ms, grads):
                m.mul_(beta).add_(g, alpha=1 - beta)
        else:
            self.ms, self._counter_m = None, 0
        if self._uniform() < self.preconditioner_update_probability:                                  # psgd.py:615-618
            first, last = self.update_preconditioner_first, not self.update_preconditioner_first
        else:
            first, last = False, False
        upd = update_precond_kron_whiten_quad4p if self._p4 else update_precond_kron_whiten_q0p5eq1p5
        apply_ = (lambda Q, G: apply_q_kron(Q, G)) if self._p4 else precond_grad_kron                 # psgd.py:573 / 575

        def update_all():
            for QL, kinds, x in zip(self.QLs, self.kinds, grads if self._whiten_grad else self.ms):
                upd(QL, x, self._noise_for(x, kinds), lr=self.lr_preconditioner, betaL=self.betaL, damping=self.damping)
        if first:
            update_all()                                                                              # psgd.py:620-626
        src = self.ms if self.momentum > 0 else grads                                                 # psgd.py:628-631
        pre = [apply_(QL[0], x) for QL, x in zip(self.QLs, src)]
        if last:
            update_all()                                                                              # psgd.py:633-639
        max_avg_amp, max_element_amp = self.grad_clip_max_amps                                        # psgd.py:642-651
        for p, h in zip(self.params, pre):
            avg_amp = torch.sqrt(torch.mean(h * h))
            if avg_amp > max_avg_amp:
                h = h * (max_avg_amp / avg_amp)
            h = h.clamp(min=-max_element_amp, max=max_element_amp)
            p.subtract_(h.view_as(p), alpha=self.lr_params)


class LRAWhitenOracle:
    """psgd.py:1075-1190 restated with explicit gradients instead of a closure (the closure/autograd front end
    is out of scope); U/V initial values and every random draw are supplied by the caller."""

    def __init__(self, params: List[Tensor], U0: Tensor, V0: Tensor, preconditioner_init_scale: Optional[float] = None,
                 lr_params=0.001, lr_preconditioner=0.1, betaL=0.9, damping=1e-9, momentum=0.0,
                 grad_clip_max_amps=(2.0, 10.0), preconditioner_update_probability=1.0,
                 update_preconditioner_first=True, whiten_grad=True):
        self.params = params
        self.lr_params, self.lr_preconditioner, self.betaL, self.damping = lr_params, lr_preconditioner, betaL, damping
        self.momentum = momentum if (0 < momentum < 1) else 0.0
        self.grad_clip_max_amps = grad_clip_max_amps
        self.preconditioner_update_probability = preconditioner_update_probability
        self.update_preconditioner_first = update_preconditioner_first
        self.whiten_grad = whiten_grad
        dtype = params[0].dtype
        self.sizes = [p.numel() for p in params]
        n = sum(self.sizes)
        self.UVd = [U0.clone(), V0.clone()]                                # psgd.py:1115-1118 (values supplied)
        if preconditioner_init_scale is not None:
            self.UVd.append(torch.ones(n, 1, dtype=dtype) * preconditioner_init_scale)
        self.Luvd = [lift2single(torch.zeros([], dtype=dtype)) for _ in range(3)]
        self.m, self.counter_m = None, 0

    @torch.no_grad()
    def step(self, grads: List[Tensor], gate_u: float, v_noise: Optional[Tensor], coin_u: Optional[float]) -> None:
        grad = torch.cat([torch.reshape(g, [-1, 1]) for g in grads])       # psgd.py:1142
        if len(self.UVd) < 3:                                               # psgd.py:1144-1145
            self.UVd.append((torch.mean(grad ** 4) + self.damping ** 4) ** (-1 / 8) * torch.ones_like(grad))
        if self.momentum > 0:                                               # psgd.py:1147-1155
            beta = min(self.counter_m / (1 + self.counter_m), self.momentum)
            self.counter_m += 1
            if self.m is None:
                self.m = torch.zeros_like(grad)
            self.m.mul_(beta).add_(grad, alpha=1 - beta)
        else:
            self.m, self.counter_m = None, 0
        if gate_u < self.preconditioner_update_probability:                # psgd.py:1157-1160
            first, last = self.update_preconditioner_first, not self.update_preconditioner_first
        else:
            first, last = False, False
        target = grad if self.whiten_grad else self.m
        if first:
            update_precond_lra_whiten(self.UVd, self.Luvd, target, v_noise, coin_u, lr=self.lr_preconditioner,
                                      betaL=self.betaL, damping=self.damping)
        pre_grad = precond_grad_lra(self.UVd, self.m if self.momentum > 0 else grad)
        if last:
            update_precond_lra_whiten(self.UVd, self.Luvd, target, v_noise, coin_u, lr=self.lr_preconditioner,
                                      betaL=self.betaL, damping=self.damping)
        max_avg_amp, max_element_amp = self.grad_clip_max_amps             # psgd.py:1179-1183
        avg_amp = torch.sqrt(torch.mean(pre_grad * pre_grad))
        if avg_amp > max_avg_amp:
            pre_grad = pre_grad * (max_avg_amp / avg_amp)
        pre_grad = pre_grad.clamp(min=-max_element_amp, max=max_element_amp)
        off = 0                                                             # psgd.py:1186-1187
        for p, n in zip(self.params, self.sizes):
            p.subtract_(pre_grad[off:off + n].view_as(p), alpha=self.lr_params)
            off += n


# --------------------------------------------------------------------------------------------
# FLOP model used by bench.py for the roofline line (SURVEY 8d / BASELINE.md section 3)
# --------------------------------------------------------------------------------------------
def kron_step_flops(shape: Sequence[int], max_size: float = float("inf"), max_skew: float = 1.0) -> Tuple[float, float]:
    """Returns (update+2*apply FLOPs, apply-only FLOPs) for one tensor; diagonal factors count 0."""
    kinds = kron_factor_kinds([s for s in shape if s != 1], max_size, max_skew)
    dims = [s for s in shape if s != 1]
    N = math.prod(dims) if dims else 1
    step = 0.0
    apply_only = 0.0
    for d, kind in zip(dims, kinds):
        if kind != "dense":
            continue
        ap = min(4.0 * N * d, 2.0 * d ** 3 + 2.0 * N * d)
        up = 2.0 * N * d + 6.0 * d ** 3 + 512.0 * d ** 2
        step += 2 * ap + up
        apply_only += ap
    return step, apply_only
