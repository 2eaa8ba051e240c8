#!/usr/bin/env python3
"""Observed parity of the HIP path against the reference's golden vectors (tests/golden), through the C ABI -- the numbers
behind the pass/fail thresholds of tests/test_gpu_*.py.  Run on an MI355X:  python tests/parity_report.py > profiles/<name>.md  (lives under tests/: it uses the oracle)
fp32: max over steps of relerr(hip, golden).  bf16: max over steps of relerr(hip, fp64 oracle) next to relerr(reference bf16,
fp64 oracle) on the same inputs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from helpers import DT, P_of, T, golden_names, kron_dtypes, kron_noise_from_golden, load, relerr  # noqa: E402
from oracle import psgd_oracle as orc  # noqa: E402
import psgd_torch_amd as amd  # noqa: E402

DEV = "cuda:0"
GEOM = {"kron": ("Q0.5EQ1.5", amd.update_precond_kron_whiten_q0p5eq1p5, orc.update_precond_kron_whiten_q0p5eq1p5, True),
        "kroneq": ("EQ", amd.update_precond_kron_whiten_eq, orc.update_precond_kron_whiten_eq, False),
        "kronqeq": ("QEQ", amd.update_precond_kron_whiten_qeq, orc.update_precond_kron_whiten_qeq, False),
        "kronquad": ("QUAD", amd.update_precond_kron_whiten_quad, orc.update_precond_kron_whiten_quad, False),
        "kronqep": ("QEP", amd.update_precond_kron_whiten_qep, orc.update_precond_kron_whiten_qep, False)}


def run(prefix):
    dq, upd_amd, upd_orc, gauge = GEOM[prefix]
    worst = {"fp32": {}, "bf16": {}}
    for name in golden_names(prefix + "_"):
        z = load(name)
        kw = dict(Scale=float(z["Scale"]), max_size=float(z["max_size"]), max_skew=float(z["max_skew"]))
        lr, betaL, damping = float(z["lr"]), float(z["betaL"]), float(z["damping"])
        for dn in kron_dtypes(z):
            if dn == "fp64":
                continue
            dt = DT[dn]
            QL, exprs = amd.init_kron(T(z["G0"], dt).to(DEV), dQ=dq, **kw)
            QL64, kinds = orc.init_kron(T(z["G0"], torch.float64), **kw)
            for t in range(int(z["T"])):
                Gd = T(z[f"G{t}"], dt)
                nz = kron_noise_from_golden(z, dn, t, len(QL[0]), dt)
                dev_noise = ([nz.g_noise.to(DEV)], {(0, i): x.to(DEV) for i, x in enumerate(nz.spd) if x is not None},
                             {(0, i): x.to(DEV) for i, x in enumerate(nz.skh) if x is not None})
                kwargs = dict(lr=lr, betaL=betaL, damping=damping, noise=dev_noise)
                if dq != "QEP":
                    kwargs["balance"] = nz.balance_u < 0.01
                upd_amd(QL, exprs, Gd.to(DEV), **kwargs)
                h = amd.precond_grad_kron(QL, exprs, Gd.to(DEV))
                n64 = orc.KronNoise(nz.g_noise.double(), [x.double() if x is not None else None for x in nz.spd],
                                    [x.double() if x is not None else None for x in nz.skh], nz.balance_u)
                upd_orc(QL64, Gd.double(), n64, lr=lr, betaL=betaL, damping=damping)
                h64 = orc.precond_grad_kron(QL64[0], Gd.double())
                items = [("h", h, z[f"{dn}_t{t}_h"], h64)]
                for i in range(len(QL[0])):
                    if gauge:
                        items.append(("P", P_of([QL[0][i]])[0], P_of([torch.from_numpy(z[f"{dn}_t{t}_Q{i}"])])[0], P_of([QL64[0][i]])[0]))
                    else:
                        items.append(("Q", QL[0][i], z[f"{dn}_t{t}_Q{i}"], QL64[0][i]))
                    items.append(("L", QL[1][i], z[f"{dn}_t{t}_L{i}"], QL64[1][i]))
                for what, got, gold, truth in items:
                    w = worst[dn].setdefault(what, [0.0, 0.0])
                    if dn == "fp32":
                        w[0] = max(w[0], relerr(got, gold) / (t + 1))
                    else:
                        w[0] = max(w[0], relerr(got, truth)); w[1] = max(w[1], relerr(gold, truth))
    return worst


def main():
    print("# Observed parity on MI355X (through the C ABI, replayed noise)\n")
    print("fp32: max over cases and steps of relerr(HIP, reference fp32) / (step index + 1)   [test bound 3e-5]")
    print("bf16: max relerr vs the fp64 oracle trajectory: HIP | reference bf16              [test bound 1.5 x reference + 2e-2 (4e-2 for L)]\n")
    print("| geometry | quantity | fp32 | bf16 HIP | bf16 reference |")
    print("|---|---|---:|---:|---:|")
    for prefix in GEOM:
        w = run(prefix)
        for what in sorted(w["fp32"]):
            b = w["bf16"].get(what, [float("nan"), float("nan")])
            print(f"| {GEOM[prefix][0]} | {what} | {w['fp32'][what][0]:.1e} | {b[0]:.1e} | {b[1]:.1e} |")


if __name__ == "__main__":
    main()
