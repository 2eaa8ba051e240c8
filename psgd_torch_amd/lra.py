"""LRA preconditioner on the HIP engine -- same names / argument meaning as the reference's psgd.py:

    update_precond_lra_whiten(UVd, Luvd, g, lr, betaL, damping)        psgd.py:1066
    precond_grad_lra(UVd, g)                                           psgd.py:1055
    LRAWhiten(params, rank_of_approximation, ...).step(closure)        psgd.py:1075-1190

U, V (N x r), d (N x 1) and the three Lipschitz scalars are ordinary torch tensors owned by the caller, updated in
place; the streaming kernels live behind psgdk_lra_* (include/psgdk.h).  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L
from .engine import _on_device


GRAM_EVERY = 16      # updates between two true Gram passes (psgdk_lra_set_gram_recurrence); 0 = psgd.py:1006 as written, every update


class _LraEngine:
    """One psgdk_lra object bound to (U, V, d, Luvd)."""

    def __init__(self, UVd, Luvd3: torch.Tensor):
        U, V, d = UVd
        if d.device.type != "cuda":
            raise L.PsgdkError(L.PSGDK_ERR_INVALID, "LRA engine needs tensors on a ROCm device; there is no CPU fallback")
        self.lib = L.lib()
        self.N, self.r = d.shape[0], U.shape[1]
        for t in (U, V, d):
            if not t.is_contiguous():
                raise L.PsgdkError(L.PSGDK_ERR_INVALID, "U, V, d must be contiguous")
        if self.r > 256:
            #  the general path (64 < r <= 1024) does its r x r work -- E @ E, the pivoted LU, two solves -- in ONE workgroup:
            # O(r^3) near-serial multiply-adds per update (~1e9 at r = 1024).  Correct, covered by goldens at 96 and 130 and by fuzz to 200; slow.
            import warnings
            warnings.warn(f"LRA rank {self.r}: above 256 the r x r stages of an update run in a single workgroup (O(r^3) per update); "
                          "expect the update to be dominated by them", RuntimeWarning, stacklevel=3)
        self.h = C.c_void_p()
        L.check(self.lib.psgdk_lra_create(C.byref(self.h), self.N, self.r, L.dtype_code(d.dtype)), "lra_create")
        wb = C.c_size_t()
        L.check(self.lib.psgdk_lra_work_bytes(self.h, C.byref(wb)), "lra_work_bytes")
        self.work = torch.zeros(wb.value, dtype=torch.uint8, device=d.device)
        self.keep = (U, V, None, Luvd3)       # (not d: the engine hangs on d itself, see _engine_for)
        self.device = d.device if d.device.index is not None else torch.device("cuda", torch.cuda.current_device())
        L.check(self.lib.psgdk_lra_bind(self.h, U.data_ptr() if self.r else None, V.data_ptr() if self.r else None, d.data_ptr(),
                                        Luvd3.data_ptr(), self.work.data_ptr()), "lra_bind")
        # The Grams of psgd.py:1006 are carried from update to update (include/psgdk.h: psgdk_lra_set_gram_recurrence) and re-read from the
        # factors every GRAM_EVERY updates.  The engine writes U and V through raw pointers, which torch's version counters do not see: a
        # counter that moved means SOMEBODY ELSE wrote the factor (a checkpoint copied in, a test poking it) -- then the Grams are re-read.
        self.gram_every = GRAM_EVERY if 0 < self.r <= 64 else 0
        if self.gram_every:
            L.check(self.lib.psgdk_lra_set_gram_recurrence(self.h, self.gram_every), "lra_set_gram_recurrence")
        self._versions = (U._version, V._version)

    def __del__(self):
        try:
            if self.h.value:
                self.lib.psgdk_lra_destroy(self.h)
                self.h = C.c_void_p()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def owns_luvd(self, Luvd) -> bool:
        base = self.keep[3]
        return all(isinstance(x, torch.Tensor) and x.data_ptr() == base[i].data_ptr() for i, x in enumerate(Luvd))

    def info(self) -> dict:
        """Which row kernels the last update / apply call took (psgdk_lra_info)."""
        out = {}
        for name, code in (("packed_rows", L.LRA_INFO_PACKED_ROWS), ("gram_age", L.LRA_INFO_GRAM_AGE)):
            v = C.c_int64()
            L.check(self.lib.psgdk_lra_info(self.h, code, C.byref(v)), "lra_info")
            out[name] = int(v.value)
        return out

    def last_sumsq_ptr(self):
        """Device word with the sum of squares of the last precond_grad output (the RMS clip reads it on the device)."""
        p = C.c_void_p()
        L.check(self.lib.psgdk_lra_last_sumsq(self.h, C.byref(p)), "lra_last_sumsq")
        return p

    @_on_device
    def update_whiten(self, g, lr, betaL, damping, v_noise=None, seed=0, offset=0, update_u=True):
        g = g.contiguous()
        vn = v_noise.to(g.dtype).contiguous() if v_noise is not None else None
        self._k = (g, vn)
        ver = (self.keep[0]._version, self.keep[1]._version)
        if ver != self._versions:
            self._versions = ver
            L.check(self.lib.psgdk_lra_state_changed(self.h), "lra_state_changed")
        L.check(self.lib.psgdk_lra_update_whiten(self.h, g.data_ptr(), vn.data_ptr() if vn is not None else None, int(seed),
                                                 int(offset), int(bool(update_u)), float(lr), float(betaL), float(damping),
                                                 self._stream()), "lra_update_whiten")

    @_on_device
    def precond_grad(self, g):
        g = g.contiguous()
        out = torch.empty_like(g)
        L.check(self.lib.psgdk_lra_precond_grad(self.h, g.data_ptr(), out.data_ptr(), self._stream()), "lra_precond_grad")
        return out

    # ---- the phase interface of a row shard (include/psgdk.h "row shards of ONE LRA preconditioner"; driven by lra_sharded.RowShardedLRA) ----
    UPDATE_PHASES, APPLY_PHASES = 5, 3

    def set_row_shard(self, row0: int):
        L.check(self.lib.psgdk_lra_set_row_shard(self.h, int(row0)), "lra_set_row_shard")

    @property
    def scratch(self) -> torch.Tensor:
        """The work buffer as fp32 words: the reduction slots psgdk_lra_phase_segments names live at its start."""
        return self.work[: (self.work.numel() // 4) * 4].view(torch.float32)

    def segments(self, kind: int, phase: int):
        n = C.c_int()
        off, cnt, op = (C.c_int64 * 4)(), (C.c_int * 4)(), (C.c_int * 4)()
        L.check(self.lib.psgdk_lra_phase_segments(self.h, int(kind), int(phase), C.byref(n), off, cnt, op), "lra_phase_segments")
        return [(int(off[i]), int(cnt[i]), int(op[i])) for i in range(n.value)]

    @_on_device
    def update_phase(self, phase, g, lr, betaL, damping, v_noise=None, seed=0, offset=0, update_u=True):
        if phase == 0:
            g = g.contiguous()
            vn = v_noise.to(g.dtype).contiguous() if v_noise is not None else None
            self._k = (g, vn)
        g, vn = self._k
        L.check(self.lib.psgdk_lra_update_phase(self.h, int(phase), g.data_ptr(), vn.data_ptr() if vn is not None else None, int(seed),
                                                int(offset), int(bool(update_u)), float(lr), float(betaL), float(damping),
                                                self._stream()), "lra_update_phase")

    @_on_device
    def apply_phase(self, phase, g, out):
        L.check(self.lib.psgdk_lra_apply_phase(self.h, int(phase), g.data_ptr(), out.data_ptr(), self._stream()), "lra_apply_phase")


def _engine_for(UVd, Luvd, rebind: bool = True) -> _LraEngine:
    """The reference passes (UVd, Luvd) lists of tensors around; the engine needs the three L scalars contiguous, so the
    first UPDATE call re-homes them into one 3-element fp32 tensor and makes Luvd[i] views of it (values preserved).
    The engine lives ON the d tensor (an attribute), so it is freed with the tensors it is bound to -- no global cache."""
    d = UVd[2]
    eng = getattr(d, "_psgdk_lra", None)
    key = (UVd[0].data_ptr() if UVd[0].numel() else 0, UVd[1].data_ptr() if UVd[1].numel() else 0)
    if eng is not None and eng.key != key:
        eng = None                        # U / V were replaced: bind afresh
    if eng is not None and rebind and not eng.owns_luvd(Luvd):
        eng = None                        # bound earlier by precond_grad_lra with placeholder scalars: take the caller's now
    if eng is None:
        L3 = torch.stack([x.detach().to(torch.float32).reshape(()) for x in Luvd]).to(d.device)
        if rebind:
            for i in range(3):
                Luvd[i] = L3[i]
        eng = _LraEngine(UVd, L3)
        eng.key = key
        d._psgdk_lra = eng
    return eng


def _check_vec(g, d):
    if g.device != d.device or g.dtype != d.dtype or g.numel() != d.numel():
        raise L.PsgdkError(L.PSGDK_ERR_INVALID, f"the vector must match d: {g.dtype}/{g.device}/{g.numel()} vs {d.dtype}/{d.device}/{d.numel()}")


def update_precond_lra_whiten(UVd, Luvd, g, lr=0.1, betaL=0.9, damping=1e-9, *, v_noise=None, coin=None):
    """psgd.py:1066-1072 (-> 994-1052), in place.  Randomness: the Philox seed and the U-or-V coin (psgd.py:1035) are
    drawn from torch's global CPU generator unless `v_noise` / `coin` are given (parity tests)."""
    _check_vec(g, UVd[2])
    eng = _engine_for(UVd, Luvd)
    seed = int(torch.randint(0, 2 ** 62, ()).item()) if v_noise is None else 0
    if coin is None:
        coin = float(torch.rand([]))
    eng.update_whiten(g, lr, betaL, damping, v_noise=v_noise, seed=seed, offset=0, update_u=coin < 0.5)


def precond_grad_lra(UVd, g):
    """psgd.py:1055-1063.  (Before any update_precond_lra_whiten on this UVd the engine binds with placeholder Lipschitz
    scalars; the first update re-binds with the caller's.)"""
    _check_vec(g, UVd[2])
    eng = getattr(UVd[2], "_psgdk_lra", None)
    if eng is None or eng.key != (UVd[0].data_ptr() if UVd[0].numel() else 0, UVd[1].data_ptr() if UVd[1].numel() else 0):
        eng = _engine_for(UVd, [torch.zeros([], dtype=torch.float32, device=UVd[2].device) for _ in range(3)], rebind=False)
    return eng.precond_grad(g)


class _FlatVectors:
    """psgdk_flat_*: all parameters of an LRAWhiten as ONE N-vector (concatenation in parameter order, psgd.py:1142)."""

    def __init__(self, params, dtype, device):
        self.lib = L.lib()
        self.device = device if device.index is not None else torch.device("cuda", torch.cuda.current_device())
        self.numels = [int(p.numel()) for p in params]
        self.n = len(params)
        self.N = sum(self.numels)
        offs, run = [], 0
        for n in self.numels:
            offs.append(run); run += n
        self._h = C.c_void_p()
        na, oa = (C.c_int64 * self.n)(*self.numels), (C.c_int64 * self.n)(*offs)
        L.check(self.lib.psgdk_flat_create(C.byref(self._h), self.n, na, oa), "flat_create")
        self.dtype = dtype
        with torch.cuda.device(self.device):
            self.g = torch.zeros(self.N, 1, dtype=dtype, device=self.device)          # the concatenated gradient
            self.sum_g4 = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._keep = None

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                self.lib.psgdk_flat_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check(self, tensors, what):
        from .engine import _check_tensors
        return _check_tensors(what, tensors, self.numels, self.device)

    @_on_device
    def gather(self, grads, m=None, beta=0.0, want_g4=False):
        grads = [g if g.is_contiguous() else g.contiguous() for g in grads]
        gdt = self._check(grads, "grads")
        ga = L.ptr_array(grads)
        self._keep = (ga, grads)
        L.check(self.lib.psgdk_flat_gather(self._h, ga, L.dtype_code(gdt), self.g.data_ptr(), L.dtype_code(self.dtype),
                                           m.data_ptr() if m is not None else None, float(beta),
                                           self.sum_g4.data_ptr() if want_g4 else None, self._stream()), "flat_gather")

    @_on_device
    def apply_clipped(self, params, h, lr, hsq_ptr, max_avg_amp, max_elem_amp):
        pdt = self._check(params, "params")
        pa = L.ptr_array(params)
        self._keep = (pa, list(params), h)
        L.check(self.lib.psgdk_flat_apply_clipped(self._h, pa, L.dtype_code(pdt), h.data_ptr(), L.dtype_code(h.dtype), float(lr), hsq_ptr,
                                                  self.N, float(max_avg_amp), float(max_elem_amp), self._stream()), "flat_apply_clipped")


class LRAWhiten:
    """The closure-style LRA optimizer of the reference (psgd.py:1075-1190): same constructor arguments, the same attributes a
    user may anneal between steps, the same step(closure) protocol -- built on the engine rather than on per-step tensor
    algebra: the parameters are ONE resident N-vector layout (no per-step torch.cat), the momentum update rides the gather
    pass, the RMS / element clipping and the scatter back into the parameters are one launch reading the device-side
    ||h||^2 that the apply pass left behind (no host synchronisation anywhere in a step)."""

    def __init__(self, params_with_grad, rank_of_approximation: int = 10, preconditioner_init_scale: Optional[float] = None,
                 lr_params=0.001, lr_preconditioner=0.1, betaL=0.9, damping=1e-9, momentum=0.0, grad_clip_max_amps=(2.0, 10.0),
                 preconditioner_update_probability=1.0, update_preconditioner_first=True, whiten_grad=True, *,
                 shard_rows: bool = False, process_group=None, seed: int = 0):
        # the reference's mutable members (psgd.py:1094-1103)
        self.lr_params, self.lr_preconditioner, self.betaL, self.damping = lr_params, lr_preconditioner, betaL, damping
        self.momentum = momentum if (0 < momentum < 1) else 0.0
        self.grad_clip_max_amps = grad_clip_max_amps
        self.preconditioner_update_probability = preconditioner_update_probability
        self.update_preconditioner_first = update_preconditioner_first
        plist = [params_with_grad] if isinstance(params_with_grad, torch.Tensor) else list(params_with_grad)
        self._params_with_grad = [p for p in plist if p.requires_grad]
        p0 = self._params_with_grad[0]
        self._vec = _FlatVectors(self._params_with_grad, p0.dtype, p0.device)
        N, r = self._vec.N, int(rank_of_approximation)
        assert 0 <= r < N, "Rank r should be in range [0, number of total parameters)"
        if r > 1024:
            raise NotImplementedError("the HIP LRA kernels hold rank <= 1024 (psgdk_lra_create: PSGDK_ERR_UNSUPPORTED); ranks above 64 take "
                                      "the general path")
        dev, dt = self._vec.device, p0.dtype
        # shard_rows=True under an initialised process group: U, V, d are cut by rows over the ranks (lra_sharded.py; SURVEY 8e, last row).
        # Gradients and parameters stay whole on every rank (the DDP surface); every rank must pass the same `seed`: the gates, the
        # U-or-V coin and the Philox seed of the damping noise come from a private generator, not from torch's global one.
        import torch.distributed as dist
        self._shard = None
        world = dist.get_world_size(process_group) if (shard_rows and dist.is_available() and dist.is_initialized()) else 1
        if shard_rows and world > 1:
            from . import lra_sharded
            rank = dist.get_rank(process_group)
            row0, rows = lra_sharded.shard_rows(N, world, rank)
            every = [lra_sharded.shard_rows(N, world, k)[1] for k in range(world)]
            if min(every) <= r:
                raise ValueError(f"shard_rows: {N} rows over {world} ranks leaves a rank with {min(every)} rows for rank-{r} factors; "
                                 "use fewer ranks or replicas")
            self._shard = dict(world=world, rank=rank, row0=row0, rows=rows, group=process_group, driver=None)
            self._gen = torch.Generator().manual_seed(int(seed))
        n_local = self._shard["rows"] if self._shard else N

        def unit_scaled(k):                                                  # psgd.py:1115-1118
            if self._shard is None:
                x = torch.randn(N, r, dtype=dt, device=dev)
                return x * (0.1 ** 0.5 / torch.linalg.vector_norm(x)) if r else x
            g_ = torch.Generator(device=dev).manual_seed(int(seed) * 1000003 + 2 * self._shard["rank"] + k)       # this rank's rows of one N x r draw
            x = torch.randn(n_local, r, dtype=dt, device=dev, generator=g_)
            ss = (x.float() ** 2).sum().reshape(1)
            tot = lra_sharded.RowShardedLRA._sum_over_ranks(ss, world, process_group)
            return x * (0.1 ** 0.5 / tot.sqrt()).to(dt) if r else x
        self._UVd = [unit_scaled(0), unit_scaled(1)]
        self._init_scale = preconditioner_init_scale
        if preconditioner_init_scale is None:
            print("FYI: Will set the preconditioner initial scale on the fly. Recommend to set it manually.")
        else:
            self._UVd.append(torch.full((n_local, 1), float(preconditioner_init_scale), dtype=dt, device=dev))
        self._Luvd = [torch.zeros([], dtype=torch.float32, device=dev) for _ in range(3)]
        self._m, self._counter_m = None, 0
        self._whiten_grad = whiten_grad
        if not whiten_grad:
            assert self.momentum > 0, "Cannot whiten momentum if the momentum setting is invalid."
        # hooks for tests that replay the reference's recorded draws
        self._uniform = (lambda: float(torch.rand([]))) if self._shard is None else (lambda: float(torch.rand([], generator=self._gen)))
        self._v_noise = None

    def _update(self, target):
        if self._shard is not None:
            sh = self._shard
            coin = self._uniform()
            seed = int(torch.randint(0, 2 ** 62, (), generator=self._gen).item())
            vn = self._v_noise() if self._v_noise is not None else None
            loc = slice(sh["row0"], sh["row0"] + sh["rows"])
            self._driver().update_whiten(target[loc], lr=self.lr_preconditioner, betaL=self.betaL, damping=self.damping,
                                         v_noise=vn[loc] if vn is not None else None, seed=seed, offset=0, update_u=coin < 0.5)
            return
        update_precond_lra_whiten(self._UVd, self._Luvd, target, lr=self.lr_preconditioner, betaL=self.betaL, damping=self.damping,
                                  v_noise=self._v_noise() if self._v_noise is not None else None, coin=self._uniform())

    def _driver(self):
        """The row shard's engine + exchange driver, made when d exists (the on-the-fly scale needs the first gradient)."""
        sh = self._shard
        if sh["driver"] is None:
            from . import lra_sharded
            L3 = torch.stack([x.detach().to(torch.float32).reshape(()) for x in self._Luvd]).to(self._UVd[2].device)
            self._Luvd = [L3[i] for i in range(3)]
            eng = _LraEngine(self._UVd, L3)
            eng.set_row_shard(sh["row0"])
            self._UVd[2]._psgdk_lra = eng
            eng.key = None
            sh["driver"] = lra_sharded.RowShardedLRA(eng, sh["group"])
        return sh["driver"]

    @torch.no_grad()
    def step(self, closure):
        with torch.enable_grad():
            out = closure()
            loss = out if isinstance(out, torch.Tensor) else out[0]
            grads = torch.autograd.grad(loss, self._params_with_grad)
        vec = self._vec
        use_m = self.momentum > 0
        if use_m:                                                            # psgd.py:1147-1153
            beta = min(self._counter_m / (1 + self._counter_m), self.momentum)
            self._counter_m += 1
            if self._m is None:
                self._m = torch.zeros_like(vec.g)
        else:                                                                # psgd.py:1154-1155
            beta, self._m, self._counter_m = 0.0, None, 0
        need_d = len(self._UVd) < 3
        vec.gather(grads, m=self._m, beta=beta, want_g4=need_d)              # g (and m) as one N-vector; sum g^4 if d is unset
        if need_d:                                                           # psgd.py:1144-1145, on the device
            scale = (vec.sum_g4 / vec.N + self.damping ** 4) ** (-1 / 8)
            self._UVd.append(scale.to(vec.g.dtype) * (torch.ones_like(vec.g) if self._shard is None else
                                                      torch.ones(self._shard["rows"], 1, dtype=vec.g.dtype, device=vec.g.device)))
        if self._uniform() < self.preconditioner_update_probability:         # psgd.py:1157-1160
            first, last = self.update_preconditioner_first, not self.update_preconditioner_first
        else:
            first, last = False, False
        target = vec.g if self._whiten_grad else self._m
        if first:
            self._update(target)
        if self._shard is not None:                                          # this rank's rows of h, then the whole h on every rank
            from . import lra_sharded
            sh = self._shard
            src = self._m if use_m else vec.g
            h_loc = self._driver().precond_grad(src[sh["row0"]:sh["row0"] + sh["rows"]])
            h = lra_sharded.all_gather_rows(h_loc, vec.N, sh["world"], sh["rank"], sh["group"])
            eng = self._driver().engine        # (its sum-of-squares word holds the sum over ALL rows after the apply's last exchange)
            max_avg_amp, max_element_amp = self.grad_clip_max_amps
            vec.apply_clipped(self._params_with_grad, h, self.lr_params, eng.last_sumsq_ptr(), max_avg_amp, max_element_amp)
            if last:
                self._update(target)
            return out
        h = precond_grad_lra(self._UVd, self._m if use_m else vec.g)         # psgd.py:1168-1171
        # The clipped parameter update goes BEFORE a trailing preconditioner update: it reads ||h||^2 from a device word of the
        # engine that produced h, and on the first step a trailing update re-binds a fresh engine (own scratch block).  The
        # update reads neither h nor the parameters (psgd.py:1172-1187), so the order does not change any result.
        eng = self._UVd[2]._psgdk_lra
        max_avg_amp, max_element_amp = self.grad_clip_max_amps               # psgd.py:1179-1187, one launch
        vec.apply_clipped(self._params_with_grad, h, self.lr_params, eng.last_sumsq_ptr(), max_avg_amp, max_element_amp)
        if last:
            self._update(target)
        return out
