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


class _LraEngine:
    """One psgdk_lra object bound to (U, V, d, Luvd)."""
    _cache = {}

    def __init__(self, UVd, Luvd3: torch.Tensor):
        U, V, d = UVd
        if d.device.type != "cuda":
            raise L.PsgdkError(L.PSGDK_ERR_INVALID, "LRA engine needs tensors on a ROCm device; there is no CPU fallback")
        self.lib = L.lib()
        self.N, self.r = d.shape[0], U.shape[1]
        for t in (U, V, d):
            if not t.is_contiguous():
                raise L.PsgdkError(L.PSGDK_ERR_INVALID, "U, V, d must be contiguous")
        self.h = C.c_void_p()
        L.check(self.lib.psgdk_lra_create(C.byref(self.h), self.N, self.r, L.dtype_code(d.dtype)), "lra_create")
        wb = C.c_size_t()
        L.check(self.lib.psgdk_lra_work_bytes(self.h, C.byref(wb)), "lra_work_bytes")
        self.work = torch.zeros(wb.value, dtype=torch.uint8, device=d.device)
        self.keep = (U, V, d, Luvd3)
        self.device = d.device if d.device.index is not None else torch.device("cuda", torch.cuda.current_device())
        L.check(self.lib.psgdk_lra_bind(self.h, U.data_ptr() if self.r else None, V.data_ptr() if self.r else None, d.data_ptr(),
                                        Luvd3.data_ptr(), self.work.data_ptr()), "lra_bind")

    def __del__(self):
        try:
            if self.h.value:
                self.lib.psgdk_lra_destroy(self.h)
                self.h = C.c_void_p()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.keep[2].device).cuda_stream)

    @_on_device
    def update_whiten(self, g, lr, betaL, damping, v_noise=None, seed=0, offset=0, update_u=True):
        g = g.contiguous()
        vn = v_noise.to(g.dtype).contiguous() if v_noise is not None else None
        self._k = (g, vn)
        L.check(self.lib.psgdk_lra_update_whiten(self.h, g.data_ptr(), vn.data_ptr() if vn is not None else None, int(seed),
                                                 int(offset), int(bool(update_u)), float(lr), float(betaL), float(damping),
                                                 self._stream()), "lra_update_whiten")

    @_on_device
    def precond_grad(self, g):
        g = g.contiguous()
        out = torch.empty_like(g)
        L.check(self.lib.psgdk_lra_precond_grad(self.h, g.data_ptr(), out.data_ptr(), self._stream()), "lra_precond_grad")
        return out


def _engine_for(UVd, Luvd) -> _LraEngine:
    """The reference passes (UVd, Luvd) lists of tensors around; the engine needs the three L scalars contiguous, so the
    first call re-homes them into one 3-element fp32 tensor and makes Luvd[i] views of it (values preserved)."""
    key = (UVd[2].data_ptr(), UVd[0].data_ptr() if UVd[0].numel() else 0)
    eng = _LraEngine._cache.get(key)
    if eng is None:
        L3 = torch.stack([x.to(torch.float32).reshape(()) for x in Luvd]).to(UVd[2].device)
        for i in range(3):
            Luvd[i] = L3[i]
        eng = _LraEngine(UVd, L3)
        _LraEngine._cache[key] = eng
        if len(_LraEngine._cache) > 64:
            _LraEngine._cache.pop(next(iter(_LraEngine._cache)))
    return eng


def update_precond_lra_whiten(UVd, Luvd, g, lr=0.1, betaL=0.9, damping=1e-9, *, v_noise=None, coin=None):
    """psgd.py:1066-1072 (-> 994-1052), in place.  Randomness: the Philox seed and the U-or-V coin (psgd.py:1035) are
    drawn from torch's global CPU generator unless `v_noise` / `coin` are given (parity tests)."""
    eng = _engine_for(UVd, Luvd)
    seed = int(torch.randint(0, 2 ** 62, ()).item()) if v_noise is None else 0
    if coin is None:
        coin = float(torch.rand([]))
    eng.update_whiten(g, lr, betaL, damping, v_noise=v_noise, seed=seed, offset=0, update_u=coin < 0.5)


def precond_grad_lra(UVd, g):
    """psgd.py:1055-1063.  (Needs a prior update_precond_lra_whiten / LRAWhiten on the same UVd for the engine binding;
    otherwise binds with zeroed Lipschitz scalars.)"""
    key = (UVd[2].data_ptr(), UVd[0].data_ptr() if UVd[0].numel() else 0)
    eng = _LraEngine._cache.get(key)
    if eng is None:
        eng = _engine_for(UVd, [torch.zeros([], dtype=torch.float32, device=UVd[2].device) for _ in range(3)])
    return eng.precond_grad(g)


class LRAWhiten:
    """psgd.py:1075-1190: same constructor arguments, mutable attributes and step(closure) protocol."""

    def __init__(self, params_with_grad, rank_of_approximation: int = 10, preconditioner_init_scale: Optional[float] = None,
                 lr_params=0.001, lr_preconditioner=0.1, betaL=0.9, damping=1e-9, momentum=0.0, grad_clip_max_amps=(2.0, 10.0),
                 preconditioner_update_probability=1.0, update_preconditioner_first=True, whiten_grad=True):
        self.lr_params = lr_params
        self.lr_preconditioner = lr_preconditioner
        self.betaL = betaL
        self.damping = damping
        self.momentum = momentum if (0 < momentum < 1) else 0.0
        self.grad_clip_max_amps = grad_clip_max_amps
        self.preconditioner_update_probability = preconditioner_update_probability
        self.update_preconditioner_first = update_preconditioner_first
        params_with_grad = [params_with_grad, ] if isinstance(params_with_grad, torch.Tensor) else params_with_grad
        self._params_with_grad = [p for p in params_with_grad if p.requires_grad]
        dtype, device = self._params_with_grad[0].dtype, self._params_with_grad[0].device
        self._param_sizes = [torch.numel(p) for p in self._params_with_grad]
        self._param_cumsizes = torch.cumsum(torch.tensor(self._param_sizes), 0)
        num_params = int(self._param_cumsizes[-1])
        assert 0 <= rank_of_approximation < num_params, "Rank r should be in range [0, number of total parameters)"
        assert rank_of_approximation <= 16, "the HIP LRA kernels hold r <= 16"
        self._UVd = []
        U = torch.randn(num_params, rank_of_approximation, dtype=dtype, device=device)      # psgd.py:1115-1118
        self._UVd.append(U * (0.1 ** 0.5 / torch.linalg.vector_norm(U)) if rank_of_approximation else U)
        V = torch.randn(num_params, rank_of_approximation, dtype=dtype, device=device)
        self._UVd.append(V * (0.1 ** 0.5 / torch.linalg.vector_norm(V)) if rank_of_approximation else V)
        if preconditioner_init_scale is None:
            print("FYI: Will set the preconditioner initial scale on the fly. Recommend to set it manually.")
        else:
            self._UVd.append(torch.ones(num_params, 1, dtype=dtype, device=device) * preconditioner_init_scale)
        self._Luvd = [torch.zeros([], dtype=torch.float32, device=device) for _ in range(3)]
        self._m, self._counter_m = None, 0
        self._whiten_grad = whiten_grad
        if not whiten_grad:
            assert self.momentum > 0, "Cannot whiten momentum if the momentum setting is invalid."
        # test hooks: replay recorded draws
        self._uniform = lambda: float(torch.rand([]))
        self._v_noise = None

    @torch.no_grad()
    def step(self, closure):
        with torch.enable_grad():
            closure_returns = closure()
            loss = closure_returns if isinstance(closure_returns, torch.Tensor) else closure_returns[0]
            grads = torch.autograd.grad(loss, self._params_with_grad)
        grad = torch.cat([torch.reshape(g, [-1, 1]) for g in grads])                          # psgd.py:1142
        if len(self._UVd) < 3:                                                                  # psgd.py:1144-1145
            self._UVd.append((torch.mean(grad ** 4) + self.damping ** 4) ** (-1 / 8) * torch.ones_like(grad))
        if self.momentum > 0:                                                                   # psgd.py:1147-1155
            beta = min(self._counter_m / (1 + self._counter_m), self.momentum)
            self._counter_m += 1
            if self._m is None:
                self._m = torch.zeros_like(grad)
            self._m.mul_(beta).add_(grad, alpha=1 - beta)
        else:
            self._m, self._counter_m = None, 0
        if self._uniform() < self.preconditioner_update_probability:                            # psgd.py:1157-1160
            first, last = self.update_preconditioner_first, not self.update_preconditioner_first
        else:
            first, last = False, False
        target = grad if self._whiten_grad else self._m

        def do_update():
            vn = self._v_noise() if self._v_noise is not None else None
            update_precond_lra_whiten(self._UVd, self._Luvd, target, lr=self.lr_preconditioner, betaL=self.betaL,
                                      damping=self.damping, v_noise=vn, coin=self._uniform())
        if first:
            do_update()
        pre_grad = precond_grad_lra(self._UVd, self._m if self.momentum > 0 else grad)          # psgd.py:1168-1171
        if last:
            do_update()
        max_avg_amp, max_element_amp = self.grad_clip_max_amps                                  # psgd.py:1179-1183
        avg_amp = torch.sqrt(torch.mean(pre_grad * pre_grad))
        pre_grad = pre_grad * torch.clamp(max_avg_amp / avg_amp, max=1.0)                       # branch-free: no host sync
        pre_grad.clamp_(min=-max_element_amp, max=max_element_amp)
        for (param, i, j) in zip(self._params_with_grad, self._param_sizes, self._param_cumsizes):
            param.subtract_(pre_grad[j - i:j].view_as(param), alpha=self.lr_params)
        return closure_returns
