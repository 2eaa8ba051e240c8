"""Shared helpers for the test-suite (golden loading, error metrics, oracle replay)."""
import glob
import os

import numpy as np
import torch

from oracle import psgd_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DT = {"fp64": torch.float64, "fp32": torch.float32, "bf16": torch.bfloat16}


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def T(x, dtype):
    return torch.from_numpy(np.asarray(x)).to(dtype)


def relerr(a, b):
    """relative Frobenius error of a vs reference b, computed in fp64."""
    a = torch.as_tensor(a).detach().cpu().to(torch.float64)
    b = torch.as_tensor(b).detach().cpu().to(torch.float64)
    den = float(torch.linalg.vector_norm(b))
    num = float(torch.linalg.vector_norm(a - b))
    return num / den if den > 0 else num


def kron_dtypes(z):
    return [dn for dn in DT if f"{dn}_t0_h" in z.files]


def kron_noise_from_golden(z, dn, t, nfac, dtype):
    spd = [T(z[f"{dn}_t{t}_spd{i}"], dtype) if f"{dn}_t{t}_spd{i}" in z.files else None for i in range(nfac)]
    skh = [T(z[f"{dn}_t{t}_skh{i}"], dtype) if f"{dn}_t{t}_skh{i}" in z.files else None for i in range(nfac)]
    return orc.KronNoise(T(z[f"{dn}_t{t}_gnoise"], dtype), spd, skh, float(z[f"{dn}_t{t}_balance_u"]))


def P_of(Q):
    """P factors Q^T Q (dense) / q*q (diag) in fp64 -- the gauge-invariant quantity parity is asserted on."""
    out = []
    for q in Q:
        q = torch.as_tensor(q).detach().cpu().to(torch.float64)
        out.append(q.t() @ q if q.dim() == 2 else q * q)
    return out


def pro_noise_from_golden(z, dn, t, nfac, dtype):
    """PRO4P goldens: per dense factor the draws of its successive procrustes_step3 calls (psgd.py:447-450)."""
    out = []
    for i in range(nfac):
        key = f"{dn}_t{t}_npro{i}"
        out.append([T(z[f"{dn}_t{t}_pro{i}_{k}"], dtype) for k in range(int(z[key]))] if key in z.files else None)
    return out
