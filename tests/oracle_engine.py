"""TEST-ONLY stand-in for psgd_torch_amd.engine.KronEngine, computing with the CPU oracle.

It lets the host logic of psgd_torch_amd.KWNS4 (gating, ownership, the all-gather exchange of the sharded path) be
exercised on CPU with the gloo backend.  It lives under tests/ and is injected through KWNS4's `engine_factory`
argument; the product never constructs it (the product path fails loudly without the HIP library + a GPU).
"""
import hashlib

import torch

from oracle import psgd_oracle as orc

SRC_EMA, SRC_GRAD = 0, 1


class _TorchFlatApply:
    """TEST-ONLY counterpart of psgd_torch_amd.engine.FlatApply on CPU tensors."""

    def __init__(self, numels, offsets, device):
        self.numels, self.offsets = list(numels), list(offsets)

    def apply(self, params, flat, lr, decoupled_wd):
        for p, n, o in zip(params, self.numels, self.offsets):
            if p is None:
                continue
            if decoupled_wd:
                p.mul_(1.0 - decoupled_wd * lr)
            p.subtract_(flat[o:o + n].view_as(p).to(p.dtype), alpha=lr)


class OracleEngine:
    FlatApply = _TorchFlatApply

    def __init__(self, shapes, device, precond_dtype=torch.bfloat16, max_size=float("inf"), max_skew=1.0,
                 use_momentum=True, init_scale=1.0, tensor_ids=None, geometry="Q0.5EQ1.5"):
        self.geometry = geometry
        self._update = {"Q0.5EQ1.5": orc.update_precond_kron_whiten_q0p5eq1p5, "EQ": orc.update_precond_kron_whiten_eq,
                        "QEQ": orc.update_precond_kron_whiten_qeq, "QUAD": orc.update_precond_kron_whiten_quad,
                        "QEP": orc.update_precond_kron_whiten_qep, "QUAD4P": orc.update_precond_kron_whiten_quad4p}[geometry]
        self._apply = orc.precond_grad_kron_4p if geometry == "QUAD4P" else orc.precond_grad_kron
        self.shapes = [tuple(s) for s in shapes]
        self.n = len(self.shapes)
        self.dtype = precond_dtype
        self.use_momentum = use_momentum
        self.ids = list(tensor_ids) if tensor_ids is not None else list(range(self.n))
        self.QLs, self.kinds = [], []
        for s in self.shapes:
            ql, kinds = orc.init_kron(torch.zeros(s, dtype=precond_dtype), Scale=init_scale, max_size=max_size, max_skew=max_skew)
            self.QLs.append(ql)
            self.kinds.append(kinds)
        self.ema = [torch.zeros(s, dtype=precond_dtype) if use_momentum else None for s in self.shapes]
        self.gc = [None] * self.n
        self.h = [None] * self.n
        # like the real engine, all persistent state (Q, L, ema) lives in ONE byte arena that the tensors above are views of,
        # so that checkpoint / resync code paths that copy or broadcast `state_arena` are exercised on CPU as well
        def nbytes(t):
            return (t.numel() * t.element_size() + 15) // 16 * 16
        tensors = [t for ql in self.QLs for part in ql for t in part] + [e for e in self.ema if e is not None]
        self.state_arena = torch.zeros(max(sum(nbytes(t) for t in tensors), 16), dtype=torch.uint8)
        off = 0

        def rehome(t):
            nonlocal off
            v = self.state_arena[off:off + t.numel() * t.element_size()].view(t.dtype).view(t.shape)
            v.copy_(t)
            off += nbytes(t)
            return v
        for ql in self.QLs:
            for part in ql:
                for i in range(len(part)):
                    part[i] = rehome(part[i])
        self.ema = [rehome(e) if e is not None else None for e in self.ema]

    def QL(self, k):
        return self.QLs[k]

    def accumulate(self, grads, params=None, coupled_wd=0.0, beta=0.0, keep_grad=False, damp=None):
        for k, g in enumerate(grads):
            if coupled_wd:
                g = g.add(params[k], alpha=coupled_wd)
            g = g.squeeze().to(self.dtype)
            self.gc[k] = g
            if self.use_momentum:
                self.ema[k].mul_(beta).add_(g, alpha=1.0 - beta)

    def _src(self, source, k):
        return self.gc[k] if source == SRC_GRAD else self.ema[k]

    def _gen(self, seed, offset, tid):
        hsh = hashlib.sha256(f"{seed}:{offset}:{tid}".encode()).digest()
        return torch.Generator().manual_seed(int.from_bytes(hsh[:7], "little"))

    def update_precond(self, source, lr, betaL, damping, seed=0, offset=0, noise=None, balance_mask=None):
        assert noise is None
        for k in range(self.n):
            G = self._src(source, k)
            nz = orc.KronNoise.draw(G, self.kinds[k], self._gen(seed, offset, self.ids[k]))
            nz.balance_u = 0.0 if (balance_mask is not None and balance_mask[k]) else 1.0
            self._update(self.QLs[k], G, nz, lr=lr, betaL=betaL, damping=damping)

    def precond_grad(self, source):
        for k in range(self.n):
            self.h[k] = self._apply(self.QLs[k][0], self._src(source, k))

    def _clipped(self, k, max_avg_amp, max_elem_amp):
        h = self.h[k]
        avg = torch.sqrt(torch.mean(h * h))
        if avg > max_avg_amp:
            h = h * (max_avg_amp / avg)
        return h.clamp(min=-max_elem_amp, max=max_elem_amp)

    def apply_update(self, params, lr, decoupled_wd, max_avg_amp, max_elem_amp):
        for k, p in enumerate(params):
            if decoupled_wd:
                p.mul_(1.0 - decoupled_wd * lr)
            p.subtract_(self._clipped(k, max_avg_amp, max_elem_amp).view_as(p).to(p.dtype), alpha=lr)

    def read_precond_grad(self, k, out=None, clip=False, max_avg_amp=2.0, max_elem_amp=10.0):
        h = self._clipped(k, max_avg_amp, max_elem_amp) if clip else self.h[k]
        if out is None:
            return h.clone()
        out.copy_(h.reshape(out.shape))
        return out

    def export_precond_grad(self, outs, clip=True, max_avg_amp=2.0, max_elem_amp=10.0):
        for k, out in enumerate(outs):
            self.read_precond_grad(k, out=out, clip=clip, max_avg_amp=max_avg_amp, max_elem_amp=max_elem_amp)

    def state_changed(self):
        pass
