"""KronEngine -- thin host object around one psgdk plan (include/psgdk.h).

Owns the two arenas (as torch uint8 tensors, so that state is ordinary torch memory: checkpointable, visible to
torch.distributed) and exposes Q / L / ema of every tensor as strided torch VIEWS of the state arena, in the layout
the reference keeps in optimizer.state[p] (wrapped_as_torch_optimizer_for_ddp.py:129-137).  Every compute call goes
to the HIP library; nothing here has a CPU or eager-torch fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch

from . import _lib as L


def _on_device(fn):
    """Run an engine call with the engine's GPU current: the library launches on the stream it is handed, and HIP requires
    that stream's device to be the calling thread's current device (torch ops follow their tensors; raw launches do not)."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *a, **k):
        if torch.cuda.current_device() == self.device.index:
            return fn(self, *a, **k)
        with torch.cuda.device(self.device):
            return fn(self, *a, **k)
    return wrapper


def _numel(shape) -> int:
    n = 1
    for x in shape:
        n *= int(x)
    return n


def _explain_bad_tensor(what, tensors, numels, device, dt):
    for k, (t, n) in enumerate(zip(tensors, numels)):
        if not isinstance(t, torch.Tensor):
            raise L.PsgdkError(L.PSGDK_ERR_INVALID, f"{what}[{k}] is not a tensor")
        if t.device != device:
            raise L.PsgdkError(L.PSGDK_ERR_INVALID, f"{what}[{k}] lives on {t.device}, the engine on {device}")
        if t.dtype != dt:
            raise L.PsgdkError(L.PSGDK_ERR_INVALID, f"{what}[{k}] has dtype {t.dtype} but {what}[0] has {dt}: one engine call "
                                                    "takes one element type (bucket the tensors by dtype, as KWNS4 does)")
        if t.numel() != n:
            raise L.PsgdkError(L.PSGDK_ERR_INVALID, f"{what}[{k}] has {t.numel()} elements, the plan expects {n}")
        if not t.is_contiguous():
            raise L.PsgdkError(L.PSGDK_ERR_INVALID, f"{what}[{k}] is not contiguous (strides {tuple(t.stride())} for shape "
                                                    f"{tuple(t.shape)}): the HIP engine addresses tensors by raw pointer in "
                                                    "logical-contiguous order")
    raise L.PsgdkError(L.PSGDK_ERR_INVALID, f"{what}: a tensor failed validation")      # (not reached)


def _check_tensors(what: str, tensors: Sequence[torch.Tensor], numels: Sequence[int], device) -> torch.dtype:
    """The library takes raw device pointers and walks them in logical-contiguous order with ONE element type per call: a
    strided view (channels_last weight, transposed / tied view), a tensor on another device or a mixed-dtype list would be
    read and written wrongly without any error -- the reference's `p.subtract_(h.view_as(p))` is stride-safe
    (wrapped_as_torch_optimizer_for_ddp.py:157), raw pointers are not.  Refuse instead of corrupting."""
    if len(tensors) != len(numels):
        raise L.PsgdkError(L.PSGDK_ERR_INVALID, f"{what}: expected {len(numels)} tensors, got {len(tensors)}")
    dt = tensors[0].dtype if len(tensors) else torch.float32
    Tensor = torch.Tensor
    for t, n in zip(tensors, numels):
        # one combined test per tensor on the way every step takes (this runs for every gradient and parameter of every call);
        # what exactly is wrong is worked out only when something is
        if not (isinstance(t, Tensor) and t.dtype is dt and t.numel() == n and t.is_contiguous() and t.device == device):
            _explain_bad_tensor(what, tensors, numels, device, dt)
    L.dtype_code(dt)       # bf16 / fp32 only
    return dt


class FlatApply:
    """psgdk_flat_*: the parameter update of the sharded path's exchange step (all tensors, one launch)."""

    def __init__(self, numels: Sequence[int], offsets: Sequence[int], device):
        self.lib = L.lib()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.PsgdkError(L.PSGDK_ERR_INVALID, "FlatApply needs a ROCm device (cuda:N); there is no CPU fallback")
        self.n = len(numels)
        self.numels = [int(x) for x in numels]
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._h = C.c_void_p()
        na = (C.c_int64 * self.n)(*[int(x) for x in numels])
        oa = (C.c_int64 * self.n)(*[int(x) for x in offsets])
        L.check(self.lib.psgdk_flat_create(C.byref(self._h), self.n, na, oa), "flat_create")
        self._keep = None

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                self.lib.psgdk_flat_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def set_clip_groups(self, piece_group: Sequence[int], sum_off_bytes: Sequence[int], member_stride_bytes: int, members: int,
                        numel_clip: Sequence[int]):
        """Row-split tensors whose RMS clip is deferred to apply() (psgdk_flat_set_clip_groups): piece_group[t] = the group of piece t or -1;
        member m's fp32 partial sum of h^2 of group g lies at byte sum_off_bytes[g] + m * member_stride_bytes of the gathered buffer."""
        n_groups = len(sum_off_bytes)
        pg = (C.c_int32 * self.n)(*[int(x) for x in piece_group])
        so = (C.c_int64 * max(n_groups, 1))(*[int(x) for x in sum_off_bytes])
        nc = (C.c_int64 * max(n_groups, 1))(*[int(x) for x in numel_clip])
        L.check(self.lib.psgdk_flat_set_clip_groups(self._h, n_groups, pg, so, int(member_stride_bytes), int(members), nc), "flat_set_clip_groups")
        self.n_groups = n_groups

    def apply(self, params: Sequence[torch.Tensor], flat: torch.Tensor, lr: float, decoupled_wd: float, clip=None):
        """clip = (max_avg_amp, max_elem_amp): also clip the pieces of the declared groups (set_clip_groups) from the sums inside `flat`."""
        live = [(p, n) for p, n in zip(params, self.numels) if p is not None]      # None: skipped this step
        _check_tensors("params", [p for p, _ in live], [n for _, n in live], self.device)
        if len(params) != self.n:
            raise L.PsgdkError(L.PSGDK_ERR_INVALID, f"params: expected {self.n} entries, got {len(params)}")
        if not flat.is_contiguous() or flat.device != self.device:
            raise L.PsgdkError(L.PSGDK_ERR_INVALID, "the gathered buffer must be a contiguous tensor on the engine's device")
        pa = L.ptr_array(params)
        self._keep = (pa, list(params), flat)
        with torch.cuda.device(self.device):
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        if not live:
            return
        if clip is not None and getattr(self, "n_groups", 0):
            L.check(self.lib.psgdk_flat_apply_groups(self._h, pa, L.dtype_code(live[0][0].dtype), flat.data_ptr(), L.dtype_code(flat.dtype),
                                                     float(lr), float(decoupled_wd), float(clip[0]), float(clip[1]), st), "flat_apply_groups")
            return
        L.check(self.lib.psgdk_flat_apply(self._h, pa, L.dtype_code(live[0][0].dtype), flat.data_ptr(), L.dtype_code(flat.dtype),
                                          float(lr), float(decoupled_wd), st), "flat_apply")


class KronEngine:
    FlatApply = FlatApply

    def __init__(self, shapes: Sequence[Sequence[int]], device, precond_dtype=torch.bfloat16, max_size=float("inf"),
                 max_skew=1.0, use_momentum=True, init_scale: Optional[float] = 1.0,
                 tensor_ids: Optional[Sequence[int]] = None, geometry: str = "Q0.5EQ1.5", row_shards=None):
        """shapes: the SQUEEZED shapes of the tensors (wrapped_as_torch_optimizer_for_ddp.py:124).
        tensor_ids: global ids for the Philox noise streams (sharded optimizers pass the un-sharded indices).
        row_shards: {k: (global_rows, row0, member, members)} -- tensor k is a row block of a larger row-sharded matrix
        (include/psgdk.h, "row shards"): its update runs as update_begin / exchange / update_finish.
        geometry: the dQ of psgd.init_kron (psgd.py:161): "Q0.5EQ1.5", "EQ" (upper-triangular Q), "QEQ", "QUAD", "QEP"."""
        codes = {"Q0.5EQ1.5": L.GEOM_Q0P5EQ1P5, "Q0p5EQ1p5": L.GEOM_Q0P5EQ1P5, "EQ": L.GEOM_EQ, "QEQ": L.GEOM_QEQ,
                 "QUAD": L.GEOM_QUAD, "QEP": L.GEOM_QEP, "QUAD4P": L.GEOM_QUAD4P, "PRO4P": L.GEOM_PRO4P}
        if geometry not in codes:
            raise NotImplementedError(f"dQ={geometry!r}: built geometries are Q0.5EQ1.5, EQ, QEQ, QUAD, QEP, QUAD4P, PRO4P")
        self.geometry = codes[geometry]
        self.lib = L.lib()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.PsgdkError(L.PSGDK_ERR_INVALID, "KronEngine needs a ROCm device (cuda:N); there is no CPU fallback")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.dtype = precond_dtype
        self.code = L.dtype_code(precond_dtype)
        self.shapes = [tuple(int(x) for x in s) for s in shapes]
        self.n = len(self.shapes)
        self.numels = [_numel(s) for s in self.shapes]
        self.use_momentum = bool(use_momentum)
        ndim = (C.c_int32 * self.n)(*[len(s) for s in self.shapes])
        flat = [d for s in self.shapes for d in s]
        dims = (C.c_int64 * max(len(flat), 1))(*flat)
        self._plan = C.c_void_p()
        L.check(self.lib.psgdk_plan_create(C.byref(self._plan), self.n, ndim, dims, float(max_size), float(max_skew),
                                           self.code, int(self.use_momentum)), "plan_create")
        if tensor_ids is not None:
            assert len(tensor_ids) == self.n
            ids = (C.c_uint32 * self.n)(*[int(i) for i in tensor_ids])
            L.check(self.lib.psgdk_plan_set_stream_ids(self._plan, ids), "set_stream_ids")
        if self.geometry != L.GEOM_Q0P5EQ1P5:
            L.check(self.lib.psgdk_plan_set_geometry(self._plan, self.geometry), "set_geometry")
        self.row_shards = dict(row_shards or {})
        self.xchg = None
        for k in sorted(self.row_shards):
            grow, row0, member, members = self.row_shards[k]
            L.check(self.lib.psgdk_plan_set_row_shard(self._plan, int(k), int(grow), int(row0), int(member), int(members)), "set_row_shard")
        sb, wb = C.c_size_t(), C.c_size_t()
        L.check(self.lib.psgdk_plan_arena_bytes(self._plan, C.byref(sb), C.byref(wb)), "arena_bytes")
        with torch.cuda.device(self.device):
            self.state_arena = torch.zeros(sb.value, dtype=torch.uint8, device=self.device)
            self.work_arena = torch.zeros(wb.value, dtype=torch.uint8, device=self.device)
            L.check(self.lib.psgdk_plan_bind(self._plan, self.state_arena.data_ptr(), self.work_arena.data_ptr()), "bind")
        self._build_views()
        if self.row_shards:
            rb = C.c_size_t()
            L.check(self.lib.psgdk_plan_exchange_bytes(self._plan, C.byref(rb)), "exchange_bytes")
            members = next(iter(self.row_shards.values()))[3]
            self.member = next(iter(self.row_shards.values()))[2]
            self.xchg_record_bytes = rb.value
            with torch.cuda.device(self.device):
                # `members` records (this member's is written by update_begin; the caller all-gathers the buffer IN PLACE)
                self.xchg = torch.zeros(members * rb.value, dtype=torch.uint8, device=self.device)
        hoff, boff = C.c_int64(), C.c_int64()
        L.check(self.lib.psgdk_plan_info(self._plan, L.INFO_HSUMSQ_OFFSET, C.byref(hoff)), "plan_info")
        L.check(self.lib.psgdk_plan_info(self._plan, L.INFO_BALNORM_OFFSET, C.byref(boff)), "plan_info")
        # device views: the tensors' sums of h^2 (one float each; row shards: the caller reduces them over the members), the balancing slots
        self.hsumsq = self.work_arena[hoff.value:hoff.value + 4 * self.n].view(torch.float32)
        self.balnorm = self.work_arena[boff.value:boff.value + 8 * self.n].view(torch.float32)
        self._keep = []        # host pointer arrays kept alive until the next call
        if init_scale is not None:
            self.init_state(init_scale)

    def __del__(self):
        try:
            if getattr(self, "_plan", None) is not None and self._plan.value:
                self.lib.psgdk_plan_destroy(self._plan)
                self._plan = C.c_void_p()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------------------------
    def _typed(self, byte_off, numel, dtype):
        esz = torch.empty(0, dtype=dtype).element_size()
        return self.state_arena[byte_off:byte_off + numel * esz].view(dtype)

    def _build_views(self):
        self.kinds: List[List[int]] = []
        self.Q: List[List[torch.Tensor]] = []
        self.Lip: List[List[torch.Tensor]] = []
        self.ema: List[Optional[torch.Tensor]] = []
        for t in range(self.n):
            nf = C.c_int()
            L.check(self.lib.psgdk_plan_num_factors(self._plan, t, C.byref(nf)), "num_factors")
            kinds, qs, ls = [], [], []
            for i in range(nf.value):
                kind, off, d, ld, loff = C.c_int(), C.c_size_t(), C.c_int64(), C.c_int64(), C.c_size_t()
                L.check(self.lib.psgdk_plan_factor_view(self._plan, t, i, C.byref(kind), C.byref(off), C.byref(d),
                                                        C.byref(ld), C.byref(loff)), "factor_view")
                kinds.append(kind.value)
                if kind.value == L.DENSE:
                    base = self._typed(off.value, ld.value * ld.value, self.dtype)
                    qs.append(torch.as_strided(base, (d.value, d.value), (ld.value, 1)))
                elif kind.value == L.SCALAR:
                    qs.append(self._typed(off.value, 1, self.dtype).view(()))
                else:
                    qs.append(self._typed(off.value, d.value, self.dtype))
                ls.append(self._typed(loff.value, 1, torch.float32).view(()))
            self.kinds.append(kinds)
            self.Q.append(qs)
            self.Lip.append(ls)
            if self.use_momentum:
                off, rows, cols, ld, tr = C.c_size_t(), C.c_int64(), C.c_int64(), C.c_int64(), C.c_int()
                L.check(self.lib.psgdk_plan_ema_view(self._plan, t, C.byref(off), C.byref(rows), C.byref(cols),
                                                     C.byref(ld), C.byref(tr)), "ema_view")
                r, c, l_ = rows.value, cols.value, ld.value
                if len(self.shapes[t]) > 2:        # N-D tensors are held as the contiguous logical array
                    numel = 1
                    for x in self.shapes[t]:
                        numel *= x
                    self.ema.append(self._typed(off.value, numel, self.dtype).view(self.shapes[t]))
                    continue
                if tr.value:
                    base = self._typed(off.value, (c - 1) * l_ + r, self.dtype)
                    v = torch.as_strided(base, (r, c), (1, l_))
                else:
                    base = self._typed(off.value, (r - 1) * l_ + c, self.dtype)
                    v = torch.as_strided(base, (r, c), (l_, 1))
                self.ema.append(v.reshape(self.shapes[t]) if len(self.shapes[t]) != 2 else v)
            else:
                self.ema.append(None)

    def QL(self, t):
        """[[Q...], [L...]] of tensor t -- same structure as psgd.init_kron's first return value (psgd.py:259-260)."""
        return [self.Q[t], self.Lip[t]]

    # ------------------------------------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @_on_device
    def init_state(self, scale: float):
        L.check(self.lib.psgdk_init_state(self._plan, float(scale), self._stream()), "init_state")

    @_on_device
    def state_changed(self):
        L.check(self.lib.psgdk_state_changed(self._plan, self._stream()), "state_changed")

    @_on_device
    def accumulate(self, grads: Sequence[torch.Tensor], params: Optional[Sequence[torch.Tensor]] = None,
                   coupled_wd: float = 0.0, beta: float = 0.0, keep_grad: bool = False, damp: Optional[dict] = None):
        """damp = dict(source, damping, seed, offset): fuse the damped input of an update_precond call that will follow
        with exactly these arguments (Philox noise only) into this pass."""
        _check_tensors("grads", grads, self.numels, self.device)
        if params is not None:
            _check_tensors("params", params, self.numels, self.device)
        ga = L.ptr_array(grads)
        pa = L.ptr_array(params) if params is not None else None
        self._keep = [ga, pa, list(grads)]
        pdt = L.dtype_code(params[0].dtype) if params is not None else L.F32
        dp = None
        if damp is not None:
            d = L.Damp(int(damp["source"]), float(damp["damping"]), None, int(damp["seed"]), int(damp["offset"]))
            self._keep.append(d)
            dp = C.byref(d)
        L.check(self.lib.psgdk_accumulate(self._plan, ga, L.dtype_code(grads[0].dtype), pa, pdt, float(coupled_wd),
                                          float(beta), int(keep_grad), dp, self._stream()), "accumulate")

    @_on_device
    def update_precond(self, source: int, lr: float, betaL: float, damping: float, seed: int = 0, offset: int = 0,
                       noise=None, balance_mask: Optional[Sequence[bool]] = None):
        """noise: None (Philox) or (g_noise list[n], spd dict{(t,i): tensor}, skh dict{(t,i): tensor}).
        Dispatches on the plan's geometry: psgd.py:394-419 (Q0.5EQ1.5), 330-336 (EQ), 367-391 (QEQ), 455-483 (QUAD);
        only Q0.5EQ1.5 reads skh."""
        nz_ptr = None
        keep = []
        if noise is not None:
            g_noise, spd, skh = noise
            gl = [x.to(self.dtype).contiguous() for x in g_noise]
            ga = L.ptr_array(gl)
            sa = (C.c_void_p * (L.MAX_DIMS * self.n))()      # PSGDK_MAX_DIMS slots per tensor
            ka = (C.c_void_p * (L.MAX_DIMS * self.n))()
            for (t, i), x in spd.items():
                x = x.to(self.dtype).contiguous(); keep.append(x); sa[t * L.MAX_DIMS + i] = x.data_ptr()
            for (t, i), x in skh.items():
                x = x.to(self.dtype).contiguous(); keep.append(x); ka[t * L.MAX_DIMS + i] = x.data_ptr()
            nz = L.Noise(C.cast(ga, C.POINTER(C.c_void_p)), C.cast(sa, C.POINTER(C.c_void_p)),
                         C.cast(ka, C.POINTER(C.c_void_p)))
            keep += [gl, ga, sa, ka, nz]
            nz_ptr = C.byref(nz)
        bm = None
        if balance_mask is not None:
            bm = (C.c_uint8 * self.n)(*[1 if b else 0 for b in balance_mask])
        if self.geometry == L.GEOM_QEP:      # balances every tensor itself, first (psgd.py:346-347): no gate argument
            self._checked_update(lambda: self.lib.psgdk_update_precond_qep(self._plan, int(source), float(lr), float(betaL),
                                                                           float(damping), nz_ptr, int(seed), int(offset), self._stream()))
            self._keep_noise = keep
            return
        fn = {L.GEOM_Q0P5EQ1P5: self.lib.psgdk_update_precond_q0p5eq1p5, L.GEOM_EQ: self.lib.psgdk_update_precond_eq,
              L.GEOM_QEQ: self.lib.psgdk_update_precond_qeq, L.GEOM_QUAD: self.lib.psgdk_update_precond_quad,
              L.GEOM_QUAD4P: self.lib.psgdk_update_precond_quad4p, L.GEOM_PRO4P: self.lib.psgdk_update_precond_pro4p}[self.geometry]
        self._checked_update(lambda: fn(self._plan, int(source), float(lr), float(betaL), float(damping), nz_ptr, int(seed),
                                        int(offset), bm, self._stream()))
        self._keep_noise = keep

    def _noise_arg(self, noise):
        if noise is None:
            return None, []
        g_noise, spd, skh = noise
        gl = [x.to(self.dtype).contiguous() for x in g_noise]
        keep = []
        ga = L.ptr_array(gl)
        sa = (C.c_void_p * (L.MAX_DIMS * self.n))()
        ka = (C.c_void_p * (L.MAX_DIMS * self.n))()
        for (t, i), x in spd.items():
            x = x.to(self.dtype).contiguous(); keep.append(x); sa[t * L.MAX_DIMS + i] = x.data_ptr()
        for (t, i), x in skh.items():
            x = x.to(self.dtype).contiguous(); keep.append(x); ka[t * L.MAX_DIMS + i] = x.data_ptr()
        nz = L.Noise(C.cast(ga, C.POINTER(C.c_void_p)), C.cast(sa, C.POINTER(C.c_void_p)), C.cast(ka, C.POINTER(C.c_void_p)))
        keep += [gl, ga, sa, ka, nz]
        return C.byref(nz), keep

    @_on_device
    def update_begin(self, source: int, lr: float, betaL: float, damping: float, seed: int = 0, offset: int = 0, noise=None):
        """First half of the update of a plan with row shards (psgd.py:402-405 + this member's partial statistics into its record of
        self.xchg).  The caller then all-gathers self.xchg in place over the members and calls update_finish with the same arguments."""
        nz_ptr, keep = self._noise_arg(noise)
        self._keep_noise = keep
        self._checked_update(lambda: self.lib.psgdk_update_precond_begin(self._plan, int(source), float(lr), float(betaL), float(damping),
                                                                         nz_ptr, int(seed), int(offset), self.xchg.data_ptr(), self._stream()))

    @_on_device
    def update_finish(self, source: int, lr: float, betaL: float, damping: float, seed: int = 0, offset: int = 0, noise=None,
                      balance_mask: Optional[Sequence[bool]] = None):
        """Second half (psgd.py:406-418).  Shards flagged in balance_mask are NOT balanced here: balance_shards() does it around the
        caller's max over the members."""
        nz_ptr, keep = self._noise_arg(noise)
        self._keep_noise = keep
        bm = None
        if balance_mask is not None:
            bm = (C.c_uint8 * self.n)(*[1 if b else 0 for b in balance_mask])
        L.check(self.lib.psgdk_update_precond_finish(self._plan, int(source), float(lr), float(betaL), float(damping), nz_ptr, int(seed),
                                                     int(offset), self.xchg.data_ptr(), bm, self._stream()), "update_finish")

    @_on_device
    def balance_shards(self, which: Sequence[int], reduce_max):
        """balance_kron_precond (psgd.py:266-275) of the row shards `which` (tensor indices of this engine): reduce_max(t) must replace
        the float32 tensor t by its maximum over the members (an all-reduce) between the two phases."""
        if not which:
            return
        m = (C.c_uint8 * self.n)(*[1 if k in set(which) else 0 for k in range(self.n)])
        L.check(self.lib.psgdk_balance_phase(self._plan, m, 0, self._stream()), "balance_phase")
        reduce_max(self.balnorm[:2 * len(which)])
        L.check(self.lib.psgdk_balance_phase(self._plan, m, 1, self._stream()), "balance_phase")

    def _checked_update(self, call):
        """PSGDK_ERR_NLB_TIMEOUT is the library reporting -- once, before enqueueing anything -- that a cooperative norm-bound
        launch of an EARLIER update gave up waiting for a sibling workgroup: the factors concerned skipped that one
        preconditioner update (state valid) and the plan now runs the multi-launch route.  Say so loudly, then repeat the call."""
        status = call()
        if status == L.PSGDK_ERR_NLB_TIMEOUT:
            import warnings
            warnings.warn("psgd_torch_amd: a cooperative norm-bound launch timed out waiting for a sibling workgroup (GPU shared "
                          "with long-running kernels?); the affected Kron factors skipped one preconditioner update and this "
                          "engine now uses the multi-launch norm-bound route", RuntimeWarning, stacklevel=3)
            status = call()
        L.check(status, "update_precond")

    @_on_device
    def precond_grad(self, source: int):
        L.check(self.lib.psgdk_precond_grad(self._plan, int(source), self._stream()), "precond_grad")

    @_on_device
    def apply_update(self, params: Sequence[torch.Tensor], lr: float, decoupled_wd: float, max_avg_amp: float,
                     max_elem_amp: float):
        _check_tensors("params", params, self.numels, self.device)
        pa = L.ptr_array(params)
        self._keep_p = [pa, list(params)]
        L.check(self.lib.psgdk_apply_update(self._plan, pa, L.dtype_code(params[0].dtype), float(lr), float(decoupled_wd),
                                            float(max_avg_amp), float(max_elem_amp), self._stream()), "apply_update")

    @_on_device
    def precond_grad_apply(self, source: int, params: Sequence[torch.Tensor], lr: float, decoupled_wd: float, max_avg_amp: float,
                           max_elem_amp: float):
        """precond_grad + apply_update as one call (..._ddp.py:150-157): the parameter update runs inside the epilogue of the apply's last
        product wherever a tensor allows it (include/psgdk.h: psgdk_precond_grad_apply); h is consumed."""
        _check_tensors("params", params, self.numels, self.device)
        pa = L.ptr_array(params)
        self._keep_p = [pa, list(params)]
        L.check(self.lib.psgdk_precond_grad_apply(self._plan, int(source), pa, L.dtype_code(params[0].dtype), float(lr), float(decoupled_wd),
                                                  float(max_avg_amp), float(max_elem_amp), self._stream()), "precond_grad_apply")

    def fuse_update(self, on: bool, stagger_ticks: int = -1):
        """test / A-B hook: False = precond_grad_apply takes the two-call route on this engine; stagger_ticks: see include/psgdk_test.h"""
        L.check(self.lib.psgdk_test_fuse_mode(self._plan, int(bool(on)), int(stagger_ticks)), "test_fuse_mode")

    @_on_device
    def read_precond_grad(self, t: int, out: Optional[torch.Tensor] = None, clip: bool = False, max_avg_amp: float = 2.0,
                          max_elem_amp: float = 10.0) -> torch.Tensor:
        if out is None:
            out = torch.empty(self.shapes[t], dtype=self.dtype, device=self.device)
        _check_tensors("out", [out], [self.numels[t]], self.device)
        L.check(self.lib.psgdk_read_precond_grad(self._plan, t, out.data_ptr(), L.dtype_code(out.dtype), int(clip),
                                                 float(max_avg_amp), float(max_elem_amp), self._stream()), "read_h")
        return out

    @_on_device
    def export_precond_grad(self, outs: Sequence[torch.Tensor], clip=True, max_avg_amp: float = 2.0, max_elem_amp: float = 10.0):
        """All tensors' (clipped) preconditioned gradients into caller buffers, ONE launch (the sharded path's export into the
        rank's segment of the all-gather buffer).  clip = 2: row shards leave unclipped (FlatApply.apply(clip=...) clips them after the exchange)."""
        dt = _check_tensors("outs", outs, self.numels, self.device)
        oa = L.ptr_array(outs)
        self._keep_o = [oa, list(outs)]
        L.check(self.lib.psgdk_export_precond_grad(self._plan, oa, L.dtype_code(dt), int(clip), float(max_avg_amp), float(max_elem_amp),
                                                   self._stream()), "export_precond_grad")

    @_on_device
    def dump_noise(self, seed: int, offset: int, pro_iter: int = -1):
        """TEST hook (psgdk_test_dump_noise, include/psgdk_test.h): what the production noise path draws for (seed, offset) -- the
        damping noise of every tensor (logical shape) and the two 32 x d start blocks of every dense factor -- in the form
        update_precond takes as `noise`, so that a test can replay the engine's own Philox draws into the oracle."""
        g = [torch.empty(s, dtype=self.dtype, device=self.device) for s in self.shapes]
        ga = L.ptr_array(g)
        sa = (C.c_void_p * (L.MAX_DIMS * self.n))()
        ka = (C.c_void_p * (L.MAX_DIMS * self.n))()
        spd, skh = {}, {}
        for t in range(self.n):
            for i, kind in enumerate(self.kinds[t]):
                if kind == L.DENSE:
                    d = self.Q[t][i].shape[0]
                    spd[(t, i)] = torch.empty(32, d, dtype=self.dtype, device=self.device)
                    skh[(t, i)] = torch.empty(32, d, dtype=self.dtype, device=self.device)
                    sa[t * L.MAX_DIMS + i] = spd[(t, i)].data_ptr()
                    ka[t * L.MAX_DIMS + i] = skh[(t, i)].data_ptr()
        L.check(self.lib.psgdk_test_dump_noise(self._plan, int(seed), int(offset), ga, sa, ka, int(pro_iter), self._stream()), "dump_noise")
        return g, spd, skh

    def info(self):
        """How the plan runs (psgdk_plan_info): cooperative norm-bound launch on / how often a timeout switched it off."""
        out = {}
        for name, code in (("nlb_coop", L.INFO_NLB_COOP), ("nlb_fallbacks", L.INFO_NLB_FALLBACKS),
                           ("dense_factors", L.INFO_DENSE_FACTORS), ("max_dense_dim", L.INFO_MAX_DENSE_DIM),
                           ("update_fused", L.INFO_UPDATE_FUSED), ("nlb_member_cols", L.INFO_NLB_MEMBER_COLS)):
            v = C.c_int64()
            L.check(self.lib.psgdk_plan_info(self._plan, code, C.byref(v)), "plan_info")
            out[name] = int(v.value)
        return out

    # live profiling of the grouped-GEMM launches (bench.py)
    def profile_enable(self, on: bool = True, calls: bool = False):
        """on: an event pair on every grouped-GEMM launch (attached to the dispatch packet: no cost on the stream); calls: also an event
        pair around every hot-path call (recorded on the stream: ~4 us each)."""
        L.check(self.lib.psgdk_profile_enable(self._plan, (1 if on else 0) | (2 if (on and calls) else 0)), "profile_enable")

    @_on_device
    def profile_read(self, reset: bool = True):
        ms, n = C.c_double(), C.c_int64()
        L.check(self.lib.psgdk_profile_read(self._plan, C.byref(ms), C.byref(n), int(reset)), "profile_read")
        return ms.value, n.value

    def profile_read_fused(self):
        """(ms, launches) of the profiled GEMM launches that carried the fused parameter update; call before a resetting profile_read"""
        ms, n = C.c_double(), C.c_int64()
        L.check(self.lib.psgdk_profile_read_fused(self._plan, C.byref(ms), C.byref(n)), "profile_read_fused")
        return ms.value, n.value

    @_on_device
    def profile_read_calls(self, reset: bool = True):
        """(ms, calls): device time between the first and the last kernel of every hot-path call while profiling was enabled"""
        ms, n = C.c_double(), C.c_int64()
        L.check(self.lib.psgdk_profile_read_calls(self._plan, C.byref(ms), C.byref(n), int(reset)), "profile_read_calls")
        return ms.value, n.value
