"""Functional seam of the Kron path, same names / argument meaning as the reference's psgd.py:

    init_kron(t, Scale, max_size, max_skew, dQ)                         psgd.py:161
    update_precond_kron_whiten_q0p5eq1p5(QL, exprs, G, lr, betaL, damping)   psgd.py:394
    update_precond_kron_whiten_eq(QL, exprs, G, lr, betaL, damping)     psgd.py:330   (init_kron(..., dQ="EQ"))
    update_precond_kron_whiten_qeq / _quad / _qep(QL, exprs, G, lr, betaL, damping)   psgd.py:367 / 455 / 339
    precond_grad_kron(QL, exprs, G)                                     psgd.py:322

so that the three lines wrapped_as_torch_optimizer_for_ddp.py:84-86 can point here.  `exprs` -- compiled einsum
expressions in the reference -- is a 1-tuple holding the HIP engine (a one-tensor plan) that owns the Q/L memory;
QL[0] / QL[1] are torch views of that memory and are updated in place exactly like the reference's.
The batched, fast use of the engine is psgd_torch_amd.KWNS4; this per-tensor form exists for drop-in use and tests.
"""
from __future__ import annotations

import torch

from . import _lib as L
from .engine import KronEngine

_SUPPORTED_DQ = {"Q0.5EQ1.5", "Q0p5EQ1p5", "EQ", "QEQ", "QUAD", "QEP", "QUAD4P", "PRO4P"}


def init_kron(t: torch.Tensor, Scale=1.0, max_size=float("inf"), max_skew=1.0, dQ="Q0.5EQ1.5"):
    """psgd.py:161-263.  Returns [[Q, L], exprs]; t must live on a ROCm device in bf16 or fp32."""
    if dQ not in _SUPPORTED_DQ:
        raise NotImplementedError(f"dQ={dQ!r}: built geometries are Q0.5EQ1.5 (the one KWNS4 uses), EQ, QEQ, QUAD, QEP, QUAD4P and PRO4P")
    if t.dim() > 26:
        raise ValueError(f"Got tensor with dim {t.dim()}; einsum runs out of letters; replace 26 with larger numbers.")
    if torch.is_complex(t):
        raise NotImplementedError("real tensors only (as wrapped_as_torch_optimizer_for_ddp.KWNS4)")
    if dQ in ("QUAD4P", "PRO4P"):           # the factors are P itself: the scale is squared (psgd.py:186-187)
        Scale = Scale ** 2
    eng = KronEngine([tuple(t.shape)], t.device, precond_dtype=t.dtype, max_size=max_size, max_skew=max_skew,
                     use_momentum=False, init_scale=float(Scale), geometry=dQ)
    return [eng.QL(0), (eng,)]


def _engine(exprs) -> KronEngine:
    eng = exprs[0]
    if not isinstance(eng, KronEngine):
        raise TypeError("exprs must come from psgd_torch_amd.kron.init_kron")
    return eng


def update_precond_kron_whiten_q0p5eq1p5(QL, exprs, G, lr=0.1, betaL=0.9, damping=1e-9, *, noise=None, balance=None):
    """psgd.py:394-419, in place on QL.  Randomness: the reference draws from torch's global generators; here the
    Philox (seed, offset) pair and the 1%-balancing gate are drawn from torch's global CPU generator, so
    torch.manual_seed() makes a run reproducible.  `noise` / `balance` override them (parity tests)."""
    eng = _engine(exprs)
    if eng.geometry != L.GEOM_Q0P5EQ1P5:
        raise ValueError('QL/exprs must come from init_kron(..., dQ="Q0.5EQ1.5")')
    eng.state_changed()                      # the caller may have written into the Q views
    eng.accumulate([G.to(eng.dtype).contiguous()], keep_grad=True)
    seed = int(torch.randint(0, 2 ** 62, ()).item()) if noise is None else 0
    if balance is None:
        balance = bool(torch.rand([]) < 0.01)
    eng.update_precond(L.SRC_GRAD, lr, betaL, damping, seed=seed, offset=0, noise=noise, balance_mask=[balance])


def update_precond_kron_whiten_eq(QL, exprs, G, lr=0.1, betaL=0.9, damping=1e-9, *, noise=None, balance=None):
    """psgd.py:330-336 -> 278-319 (the triangular geometry; QL/exprs from init_kron(..., dQ="EQ")), in place on QL.
    noise = (g_noise list, spd dict, {}) overrides the Philox draws; g_noise is the probe V of psgd.py:334."""
    eng = _engine(exprs)
    if eng.geometry != L.GEOM_EQ:
        raise ValueError('QL/exprs must come from init_kron(..., dQ="EQ")')
    eng.state_changed()
    eng.accumulate([G.to(eng.dtype).contiguous()], keep_grad=True)
    seed = int(torch.randint(0, 2 ** 62, ()).item()) if noise is None else 0
    if balance is None:
        balance = bool(torch.rand([]) < 0.01)
    eng.update_precond(L.SRC_GRAD, lr, betaL, damping, seed=seed, offset=0, noise=noise, balance_mask=[balance])


def _update_family(geom, name, QL, exprs, G, lr, betaL, damping, noise, balance):
    eng = _engine(exprs)
    if eng.geometry != geom:
        raise ValueError(f'QL/exprs must come from init_kron(..., dQ="{name}")')
    eng.state_changed()
    eng.accumulate([G.to(eng.dtype).contiguous()], keep_grad=True)
    seed = int(torch.randint(0, 2 ** 62, ()).item()) if noise is None else 0
    if balance is None:
        balance = bool(torch.rand([]) < 0.01)
    eng.update_precond(L.SRC_GRAD, lr, betaL, damping, seed=seed, offset=0, noise=noise, balance_mask=[balance])


def update_precond_kron_whiten_qeq(QL, exprs, G, lr=0.1, betaL=0.9, damping=1e-9, *, noise=None, balance=None):
    """psgd.py:367-391 (dQ = Q*E*Q; QL/exprs from init_kron(..., dQ="QEQ")), in place on QL."""
    _update_family(L.GEOM_QEQ, "QEQ", QL, exprs, G, lr, betaL, damping, noise, balance)


def update_precond_kron_whiten_quad(QL, exprs, G, lr=0.1, betaL=0.9, damping=1e-9, *, noise=None, balance=None):
    """psgd.py:455-483 (quadratic form, symmetric Q; QL/exprs from init_kron(..., dQ="QUAD")), in place on QL."""
    _update_family(L.GEOM_QUAD, "QUAD", QL, exprs, G, lr, betaL, damping, noise, balance)


def update_precond_kron_whiten_quad4p(QL, exprs, G, lr=0.1, betaL=0.9, damping=1e-9, *, noise=None, balance=None):
    """psgd.py:486-513 (fits P directly; QL/exprs from init_kron(..., dQ="QUAD4P")), in place on QL.  precond_grad_kron on
    such QL/exprs applies every factor once (what KronWhiten does for this dQ, psgd.py:573)."""
    _update_family(L.GEOM_QUAD4P, "QUAD4P", QL, exprs, G, lr, betaL, damping, noise, balance)


def update_precond_kron_whiten_pro4p(QL, exprs, G, lr=0.1, betaL=0.9, damping=1e-9, *, noise=None, balance=None):
    """psgd.py:422-452 (fits P directly, dP = P^0.5 E P, with up to ten procrustes_step3 rotations per dense factor;
    QL/exprs from init_kron(..., dQ="PRO4P")), in place on QL.  Explicit noise: the skh entry of a factor holds the draws
    of its successive rotations stacked, shape (10 * 32, d)."""
    _update_family(L.GEOM_PRO4P, "PRO4P", QL, exprs, G, lr, betaL, damping, noise, balance)


def update_precond_kron_whiten_qep(QL, exprs, G, lr=0.1, betaL=0.9, damping=1e-9, *, noise=None):
    """psgd.py:339-364 (dQ = Q*E*P; balances on every call; QL/exprs from init_kron(..., dQ="QEP")), in place on QL."""
    _update_family(L.GEOM_QEP, "QEP", QL, exprs, G, lr, betaL, damping, noise, False)


def precond_grad_kron(QL, exprs, G):
    """psgd.py:322-327."""
    eng = _engine(exprs)
    eng.state_changed()
    eng.accumulate([G.to(eng.dtype).contiguous()], keep_grad=True)
    eng.precond_grad(L.SRC_GRAD)
    return eng.read_precond_grad(0)
