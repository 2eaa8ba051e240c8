"""TEST-ONLY stand-in for the HIP engine's phase interface of a row-sharded LRA preconditioner (psgd_torch_amd/lra._LraEngine after
set_row_shard; include/psgdk.h "row shards of ONE LRA preconditioner"): the same five update phases and three apply phases, the same
reduction words between them, in plain torch on CPU -- so that the CPU suite can run psgd_torch_amd.lra_sharded.RowShardedLRA (the product's
exchange logic) over a real gloo group and compare it with the single-process oracle (oracle/psgd_oracle.py, psgd.py:994-1063).
fp32 / fp64 only (no bf16 rounding points).  Never imported by the product."""
import torch


class OracleLraPhaseEngine:
    UPDATE_PHASES, APPLY_PHASES = 5, 3

    def __init__(self, U, V, d, Luvd):
        self.U, self.V, self.d, self.Luvd = U, V, d, Luvd          # this rank's rows (updated in place); Luvd: three 0-dim tensors
        self.r = r = U.shape[1]
        dt = d.dtype
        R = max(r, 1)
        # the scratch block: named runs of words, one dtype (the HIP engine's is fp32; the oracle engine keeps the test's dtype)
        names = [("UTU", R * R), ("VTV", R * R), ("VTU", R * R), ("VTX", R), ("UTVD", R), ("ATU", R), ("BTU", R), ("ATV", R), ("BTV", R),
                 ("NA2", 1), ("NB2", 1), ("MAX1", 1), ("MAX2", 1), ("VTX2", R), ("UTY", R), ("HSQ", 1)]
        self.off, o = {}, 0
        for nm, n in names:
            self.off[nm] = (o, n); o += n
        self.scratch = torch.zeros(o, dtype=dt)
        self.local = {}          # replicated small results and this rank's N-vectors between phases

    def _w(self, nm):
        o, n = self.off[nm]
        return self.scratch[o:o + n]

    def _m(self, nm):
        return self._w(nm)[: self.r * self.r].view(self.r, self.r)

    def _v(self, nm):
        return self._w(nm)[: self.r]

    def segments(self, kind, phase):
        run = lambda a, b, op: (self.off[a][0], self.off[b][0] + self.off[b][1] - self.off[a][0], op)
        if kind == 0:
            return {0: [run("UTU", "VTU", 0)], 1: [run("VTX", "UTVD", 0)], 2: [run("ATU", "BTV", 0), run("NA2", "NB2", 0)],
                    3: [run("MAX1", "MAX2", 1)], 4: []}[phase]
        return {0: [run("VTX2", "VTX2", 0)], 1: [run("UTY", "UTY", 0)], 2: [run("HSQ", "HSQ", 0)]}[phase]

    # ---- update_precond_lra (psgd.py:994-1052) with h = g + (damping + eps |g|) v (psgd.py:1070-1072) ----
    def update_phase(self, phase, g, lr, betaL, damping, v_noise=None, seed=0, offset=0, update_u=True):
        U, V, d, r, L = self.U, self.V, self.d, self.r, self.local
        if phase == 0:
            assert v_noise is not None, "the stand-in has no Philox: pass the draw"
            self.scratch[: self.off["VTX2"][0]].zero_()
            L["v"] = v_noise.to(g.dtype)
            L["h"] = g + (damping + torch.finfo(g.dtype).eps * g.abs()) * L["v"]
            if r:
                self._m("UTU").add_(U.t() @ U); self._m("VTV").add_(V.t() @ V); self._m("VTU").add_(V.t() @ U)      # psgd.py:1006
        elif phase == 1:                                                     # rotation (psgd.py:1007-1015), replicated; then the rows
            if r:
                UtU, VtV, VtU = self._m("UTU").clone(), self._m("VTV").clone(), self._m("VTU").clone()
                trU, trV = UtU.diagonal().sum(), VtV.diagonal().sum()
                rho2 = (trU / trV) ** 0.5
                rho = rho2 ** 0.5
                E = 0.1 * (UtU / rho2 - VtV * rho2) / (trU / rho2 + trV * rho2)
                E2 = 0.5 * E @ E
                eye = torch.eye(r, dtype=E.dtype)
                Mu, Mv = (eye - (E - E2)) / rho, (eye + (E + E2)) * rho
                L["UTU2"], L["VTV2"] = Mu.t() @ UtU @ Mu, Mv.t() @ VtV @ Mv
                L["A"] = Mv.t() @ VtU @ Mu + eye                             # I + V^T U of the rotated factors (psgd.py:1020-1021)
                U.copy_(U @ Mu); V.copy_(V @ Mv)
            else:
                L["UTU2"] = L["VTV2"] = L["A"] = torch.zeros(0, 0, dtype=d.dtype)
            self._v("VTX").add_((V.t() @ (d * L["h"])).reshape(-1))          # V^T (d h)
            self._v("UTVD").add_((U.t() @ (L["v"] / d)).reshape(-1))        # U^T (v / d)
        elif phase == 2:                                                     # first solve (psgd.py:1024), then Qh, invQtv and their products
            vtx = self._v("VTX").clone().reshape(-1, 1)
            y1 = torch.linalg.solve(L["A"].t(), self._v("UTVD").clone().reshape(-1, 1)) if r else vtx
            a = d * L["h"] + U @ vtx
            b = L["v"] / d - V @ y1
            L["a"], L["b"] = a, b
            self._v("ATU").add_((a.t() @ U).reshape(-1)); self._v("BTU").add_((b.t() @ U).reshape(-1))
            self._v("ATV").add_((a.t() @ V).reshape(-1)); self._v("BTV").add_((b.t() @ V).reshape(-1))
            self._w("NA2").add_((a * a).sum()); self._w("NB2").add_((b * b).sum())
        elif phase == 3:                                                     # second solve (psgd.py:1025), Ph, invPv and the two maxima
            y2 = torch.linalg.solve(L["A"], self._v("BTV").clone().reshape(-1, 1)) if r else self._v("BTV").reshape(-1, 1)
            ph = d * (L["a"] + V @ self._v("ATU").clone().reshape(-1, 1))
            ip = (L["b"] - U @ y2) / d
            phh, vip = ph * L["h"], L["v"] * ip
            L["diff"] = phh - vip
            self._w("MAX1").copy_(torch.maximum(self._w("MAX1"), phh.abs().max().reshape(1)))
            self._w("MAX2").copy_(torch.maximum(self._w("MAX2"), vip.abs().max().reshape(1)))
        else:                                                                # psgd.py:1030-1052
            Lu, Lv, Ld = self.Luvd
            ell = (self._w("MAX1") + self._w("MAX2")).reshape(())
            Ld.copy_(torch.max(betaL * Ld + (1 - betaL) * ell, ell))
            na, nb = self._w("NA2").reshape(()).sqrt(), self._w("NB2").reshape(()).sqrt()
            G = L["VTV2"] if update_u else L["UTU2"]
            ra = (self._v("ATV") if update_u else self._v("ATU")).clone().reshape(1, -1)
            rb = (self._v("BTV") if update_u else self._v("BTU")).clone().reshape(1, -1)
            qa, qb = (ra @ G @ ra.t()).clamp_min(0).reshape(()), (rb @ G @ rb.t()).clamp_min(0).reshape(())
            ell = na * qa.sqrt() + nb * qb.sqrt()
            Lx = Lu if update_u else Lv
            Lx.copy_(torch.max(betaL * Lx + (1 - betaL) * ell, ell))
            a, b = L["a"], L["b"]
            d.sub_(lr / Ld * L["diff"] * d)                                  # psgd.py:1032
            if update_u:
                U.sub_(lr / Lx * (a @ (ra @ L["A"]) - b @ (rb @ L["A"])))    # psgd.py:1043
            else:
                V.sub_(lr / Lx * ((a + V @ ra.t()) @ ra - (b + V @ rb.t()) @ rb))    # psgd.py:1052

    # ---- precond_grad_lra (psgd.py:1055-1063) ----
    def apply_phase(self, phase, g, out):
        U, V, d = self.U, self.V, self.d
        if phase == 0:
            self.scratch[self.off["VTX2"][0]:].zero_()
            self._v("VTX2").add_((V.t() @ (d * g)).reshape(-1))
        elif phase == 1:
            self.local["y"] = d * g + U @ self._v("VTX2").clone().reshape(-1, 1)
            self._v("UTY").add_((U.t() @ self.local["y"]).reshape(-1))
        else:
            out.copy_(d * (self.local["y"] + V @ self._v("UTY").clone().reshape(-1, 1)))
            self._w("HSQ").add_((out * out).sum())
