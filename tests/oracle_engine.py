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

    def set_clip_groups(self, piece_group, sum_off_bytes, member_stride_bytes, members, numel_clip):
        self.groups = (list(piece_group), list(sum_off_bytes), int(member_stride_bytes), int(members), list(numel_clip))
        self.n_groups = len(sum_off_bytes)

    def apply(self, params, flat, lr, decoupled_wd, clip=None):
        raw = flat.view(torch.uint8)
        for t, (p, n, o) in enumerate(zip(params, self.numels, self.offsets)):
            if p is None:
                continue
            h = flat[o:o + n]
            if clip is not None and getattr(self, "n_groups", 0) and self.groups[0][t] >= 0:
                # a row-split tensor's piece: the members' partial sums of h^2 from inside the gathered buffer, in member order
                pg, so, stride, members, nc = self.groups
                g = pg[t]
                tot = torch.zeros((), dtype=torch.float32)
                for m in range(members):
                    tot = tot + raw[so[g] + m * stride:so[g] + m * stride + 4].view(torch.float32)[0]
                avg = torch.sqrt(tot / nc[g]).to(h.dtype)
                if avg > clip[0]:
                    h = h * (clip[0] / avg)
                h = h.clamp(min=-clip[1], max=clip[1])
            if decoupled_wd:
                p.mul_(1.0 - decoupled_wd * lr)
            p.subtract_(h.view_as(p).to(p.dtype), alpha=lr)


class OracleEngine:
    FlatApply = _TorchFlatApply

    def __init__(self, shapes, device, precond_dtype=torch.bfloat16, max_size=float("inf"), max_skew=1.0,
                 use_momentum=True, init_scale=1.0, tensor_ids=None, geometry="Q0.5EQ1.5", row_shards=None):
        self.geometry = geometry
        # {k: (global_rows, row0, member, members)}: tensor k is a row block of a row-split matrix (include/psgdk.h, "row shards")
        self.row_shards = dict(row_shards or {})
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
            return (t.numel() * t.element_size() + 63) // 64 * 64      # (64-byte granules: the CPU matmul picks its kernels by operand alignment)
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
        self.hsumsq = torch.zeros(self.n, dtype=torch.float32)
        self.xchg = None
        if self.row_shards:
            assert geometry in ("Q0.5EQ1.5", "Q0p5EQ1p5", "QEQ", "QUAD")
            members = next(iter(self.row_shards.values()))[3]
            self.member = next(iter(self.row_shards.values()))[2]
            # one record per member: per shard [cols x cols fp32 partial Gram][64 fp32 scalars], like the HIP engine's
            self._rec_off, off = {}, 0
            for k in sorted(self.row_shards):
                c = self.shapes[k][1]
                self._rec_off[k] = off
                off += (c * c * 4 + 256 + 255) // 256 * 256
            self.xchg_record_bytes = off
            self.xchg = torch.zeros(members * off, dtype=torch.uint8)

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

    def _noise(self, k, G, seed, offset):
        """The draws of tensor k for (seed, offset): keyed by the tensor's GLOBAL id; a row block takes its rows of the whole
        matrix's damping noise and the whole matrix's start blocks -- what one engine over the whole matrix would draw."""
        if k not in self.row_shards:
            return orc.KronNoise.draw(G, self.kinds[k], self._gen(seed, offset, self.ids[k]))
        grow, row0 = self.row_shards[k][:2]
        nz = orc.KronNoise.draw(torch.zeros(grow, G.shape[1], dtype=G.dtype), self.kinds[k], self._gen(seed, offset, self.ids[k]))
        nz.g_noise = nz.g_noise[row0:row0 + G.shape[0]]
        return nz

    def update_precond(self, source, lr, betaL, damping, seed=0, offset=0, noise=None, balance_mask=None, _skip=()):
        assert noise is None and (not self.row_shards or _skip), "a plan with row shards is updated by update_begin / update_finish"
        for k in range(self.n):
            if k in _skip:
                continue
            G = self._src(source, k)
            nz = self._noise(k, G, seed, offset)
            nz.balance_u = 0.0 if (balance_mask is not None and balance_mask[k]) else 1.0
            self._update(self.QLs[k], G, nz, lr=lr, betaL=betaL, damping=damping)

    # ---- row shards: psgd.py:394-419 cut in two around the exchange of the members' partial statistics ----------------------------
    def _rec(self, member, k):
        c = self.shapes[k][1]
        o = member * self.xchg_record_bytes + self._rec_off[k]
        return self.xchg[o:o + c * c * 4].view(torch.float32).view(c, c), self.xchg[o + c * c * 4:o + c * c * 4 + 4].view(torch.float32)

    def update_begin(self, source, lr, betaL, damping, seed=0, offset=0, noise=None):
        assert noise is None
        self._pending = {}
        for k in self.row_shards:
            Q, _ = self.QLs[k]
            G = self._src(source, k)
            nz = self._noise(k, G, seed, offset)
            damp = damping + torch.finfo(G.dtype).eps * G.abs()                       # psgd.py:402-403
            Pg = orc.precond_grad_kron(Q, G + damp * nz.g_noise.to(G.dtype))
            gram, mx = self._rec(self.member, k)
            gram.copy_(orc.gram_mode(Pg.float(), 1, True))                            # this member's share of psgd.py:405 (dense factor), fp32
            t_diag = orc.gram_mode(Pg, 0, False)                                      # the diagonal factor's term1: row-local
            mx.copy_(torch.max(t_diag).float().reshape(1))
            self._pending[k] = (t_diag, nz)

    def update_finish(self, source, lr, betaL, damping, seed=0, offset=0, noise=None, balance_mask=None):
        members = next(iter(self.row_shards.values()))[3]
        for k, (grow, row0, _, _) in self.row_shards.items():
            Q, Lq = self.QLs[k]
            t_diag, nz = self._pending[k]
            total_numel = grow * self.shapes[k][1]
            term1 = torch.zeros_like(self._rec(0, k)[0])
            for m in range(members):                                                   # member order: the same bits on every member
                term1 = term1 + self._rec(m, k)[0]
            term1 = term1.to(Q[1].dtype)
            mx = torch.max(torch.stack([self._rec(m, k)[1][0] for m in range(members)])).to(Q[0].dtype)
            quad = self.geometry == "QUAD"
            # the diagonal factor (psgd.py:406-410; QEQ :380-384, QUAD :466-471) on this member's rows, with the WHOLE matrix's maximum and
            # element count
            term2 = total_numel / grow
            ell = mx + term2
            Lq[0].copy_(torch.max(betaL * Lq[0] + (1 - betaL) * ell, ell))
            if quad:
                gain = 1 - lr / 2 / Lq[0] * (t_diag - term2)
                Q[0].mul_(gain * gain)
            else:
                Q[0].mul_(1 - lr / Lq[0] * (t_diag - term2))
            # the dense factor (psgd.py:411-416; QEQ :385-388, QUAD :472-481), replicated: every member computes the same
            term2 = total_numel / Q[1].shape[0]
            ell = orc.norm_lower_bound_spd(term1, nz.spd[1]) + term2
            Lq[1].copy_(torch.max(betaL * Lq[1] + (1 - betaL) * ell, ell))
            if self.geometry == "QEQ":
                Q[1].sub_(lr / Lq[1] * (Q[1] @ term1 - Q[1] * term2))
            elif quad:
                p_ = Q[1] - lr / 2 / Lq[1] * (term1 @ Q[1] - term2 * Q[1])
                p_ = p_ - lr / 2 / Lq[1] * (p_ @ term1 - p_ * term2)
                Q[1].copy_((p_ + p_.t()) / 2)
            else:
                Q[1].sub_(lr / Lq[1] * (term1 @ Q[1] - term2 * Q[1]))
                orc.procrustes_step2(Q[1], nz.skh[1])
        self._pending = {}
        self.update_precond(source, lr, betaL, damping, seed=seed, offset=offset, balance_mask=balance_mask, _skip=set(self.row_shards))

    def balance_shards(self, which, reduce_max):
        """psgd.py:266-275 for row blocks: max |q| of the diagonal factor over ALL members' rows."""
        if not which:
            return
        norms = torch.stack([torch.max(torch.abs(q)).float() for k in which for q in self.QLs[k][0]])
        reduce_max(norms)
        for j, k in enumerate(which):
            Q = self.QLs[k][0]
            n0, n1 = norms[2 * j].to(Q[0].dtype), norms[2 * j + 1].to(Q[1].dtype)
            gmean = torch.prod(torch.stack([n0, n1])) ** (1 / 2)
            Q[0].mul_(gmean / n0)
            Q[1].mul_(gmean / n1)

    def precond_grad(self, source):
        for k in range(self.n):
            self.h[k] = self._apply(self.QLs[k][0], self._src(source, k))
            if k in self.row_shards:      # this member's share of sum h^2 (the caller sums it over the members before the clip)
                self.hsumsq[k] = torch.sum(self.h[k].float() ** 2)

    def _clipped(self, k, max_avg_amp, max_elem_amp):
        h = self.h[k]
        if k in self.row_shards:      # ..._ddp.py:153-155 over the WHOLE matrix
            avg = torch.sqrt(self.hsumsq[k] / (self.row_shards[k][0] * self.shapes[k][1])).to(h.dtype)
        else:
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
        if clip == 2 and k in self.row_shards:      # deferred: the flat apply clips this block after the exchange
            clip = False
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
