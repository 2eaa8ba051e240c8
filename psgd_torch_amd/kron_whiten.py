"""KronWhiten -- the closure-style shell of the reference (psgd.py:516-654) on the batched HIP engine.

Same constructor arguments, the same MUTABLE attributes the reference's users anneal mid-training (lr_params,
lr_preconditioner, betaL, damping, momentum, grad_clip_max_amps, preconditioner_update_probability,
update_preconditioner_first; cf. misc/gpt2.py:440, misc/vit.py:362-363) and the same step(closure) protocol, so that
scripts such as misc/gpt2.py / misc/vit.py / rnn_xor_problem_general_purpose_preconditioner.py can switch by changing
the import.  Built geometries: dQ = "Q0.5EQ1.5" (the reference's default and recommended choice, psgd.py:10-12),
"QEQ", "QUAD", "QEP", and "EQ" (the triangular one, psgd.py:278-336).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib as L
from .engine import KronEngine


class KronWhiten:
    def __init__(self, params_with_grad, preconditioner_max_size=float("inf"), preconditioner_max_skew=1.0,
                 preconditioner_init_scale: Optional[float] = None, lr_params=0.001, lr_preconditioner=0.1, betaL=0.9,
                 damping=1e-9, momentum=0.0, grad_clip_max_amps=(2.0, 10.0), preconditioner_update_probability=1.0,
                 update_preconditioner_first=True, whiten_grad=True, dQ="Q0.5EQ1.5", *, seed: int = 0):
        # mutable members (psgd.py:541-549)
        self.lr_params = lr_params
        self.lr_preconditioner = lr_preconditioner
        self.betaL = betaL
        self.damping = damping
        self.momentum = momentum if (0 < momentum < 1) else 0.0
        self.grad_clip_max_amps = grad_clip_max_amps
        self.preconditioner_update_probability = preconditioner_update_probability
        self.update_preconditioner_first = update_preconditioner_first
        # protected members
        if dQ not in {"Q0.5EQ1.5", "Q0p5EQ1p5", "EQ", "QEQ", "QUAD", "QEP", "QUAD4P", "PRO4P"}:
            raise NotImplementedError(f"dQ={dQ!r}: all seven geometries of the reference are built")
        self._dQ = dQ
        self._preconditioner_max_size = preconditioner_max_size
        self._preconditioner_max_skew = preconditioner_max_skew
        params_with_grad = [params_with_grad, ] if isinstance(params_with_grad, torch.Tensor) else params_with_grad
        self._params_with_grad = [p for p in params_with_grad if p.requires_grad]
        self._init_scale = preconditioner_init_scale
        if preconditioner_init_scale is None:
            print("FYI: Will set the preconditioner initial scale on the fly. Recommend to set it manually.")
        self._engines = None          # [(KronEngine, [indices into _params_with_grad])]: one engine per (param dtype, grad dtype)
        self._counter_m = 0
        self._step = 0
        self._whiten_grad = whiten_grad
        if not whiten_grad:
            assert self.momentum > 0, "Cannot whiten momentum if the momentum setting is invalid."
        self._seed = int(seed)
        self._gate_gen = torch.Generator().manual_seed(self._seed)
        self._replay = None           # tests: callable(indices) -> dict(noise=..., balance_mask=...) feeding recorded draws

    def _uniform(self) -> float:
        return float(torch.rand([], generator=self._gate_gen))

    @property
    def _engine(self) -> Optional[KronEngine]:
        """The engine of the first (usually only) dtype bucket."""
        return self._engines[0][0] if self._engines else None

    def _make_engines(self, grads, scale):
        # one batched engine per (parameter dtype, gradient dtype): an engine call takes ONE element type for all its tensors
        # (the reference initialises each tensor's factors in that tensor's own dtype, psgd.py:558,602)
        groups = {}
        for i, (p, g) in enumerate(zip(self._params_with_grad, grads)):
            groups.setdefault((p.dtype, g.dtype, p.device), []).append(i)
        self._engines = []
        self._QLs = [None] * len(grads)
        for (pdt, gdt, dev), idx in groups.items():
            eng = KronEngine([tuple(grads[i].shape) for i in idx], dev, precond_dtype=gdt,
                             max_size=self._preconditioner_max_size, max_skew=self._preconditioner_max_skew,
                             use_momentum=True, init_scale=float(scale) ** (2 if self._dQ in ("QUAD4P", "PRO4P") else 1),
                             geometry=self._dQ, tensor_ids=idx)
            self._engines.append((eng, idx))
            for k, i in enumerate(idx):
                self._QLs[i] = eng.QL(k)

    @torch.no_grad()
    def step(self, closure):
        with torch.enable_grad():
            closure_returns = closure()
            loss = closure_returns if isinstance(closure_returns, torch.Tensor) else closure_returns[0]
            grads = [g.squeeze().contiguous() for g in torch.autograd.grad(loss, self._params_with_grad)]   # psgd.py:594-597
        if self._engines is None:
            if self._init_scale is None:                                                                     # psgd.py:599-602
                scale = max([torch.mean((torch.abs(g)) ** 4) for g in grads])
                scale = float((scale + self.damping ** 4) ** (-1 / 8))
            else:
                scale = self._init_scale
            self._make_engines(grads, scale)
        if self.momentum > 0:                                                                                # psgd.py:604-613
            beta = min(self._counter_m / (1 + self._counter_m), self.momentum)
            self._counter_m += 1
        else:
            if self._counter_m:
                for eng, _ in self._engines:
                    for e in eng.ema:
                        e.zero_()
            beta, self._counter_m = 0.0, 0
        if self._uniform() < self.preconditioner_update_probability:                                         # psgd.py:615-618
            first, last = self.update_preconditioner_first, not self.update_preconditioner_first
        else:
            first, last = False, False
        use_m = self.momentum > 0
        src_w = L.SRC_GRAD if (self._whiten_grad or not use_m) else L.SRC_EMA
        src_p = L.SRC_EMA if use_m else L.SRC_GRAD
        t = self._step
        damp = None
        if (first or last) and self._dQ != "EQ" and self._replay is None:      # (the EQ update builds its own (V, Hvp) pair)
            damp = dict(source=src_w, damping=self.damping, seed=self._seed, offset=2 * t + (0 if first else 1))
        for eng, idx in self._engines:
            g_ = [grads[i] for i in idx]
            if use_m:
                eng.accumulate(g_, beta=beta, keep_grad=(src_w == L.SRC_GRAD), damp=damp)
            else:
                # no momentum: the EMA buffer is bypassed (beta = 0 would overwrite it with the gradient, which is harmless
                # because a later switch to momentum > 0 restarts from counter 0 exactly like psgd.py:612-613)
                eng.accumulate(g_, beta=0.0, keep_grad=True, damp=damp)

        def update_all(offset):
            # psgd.py:620-626 / 633-639: every tensor's update (with its own draws, in parameter order) before anything else
            for eng, idx in self._engines:
                if self._replay is not None:
                    draws = self._replay(idx)
                else:
                    draws = dict(noise=None, balance_mask=[self._uniform() < 0.01 for _ in idx])
                eng.update_precond(src_w, self.lr_preconditioner, self.betaL, self.damping, seed=self._seed, offset=offset, **draws)
        if first:
            update_all(2 * t)
        max_avg_amp, max_element_amp = self.grad_clip_max_amps                                                # psgd.py:642-651
        for eng, idx in self._engines:
            eng.precond_grad(src_p)
            eng.apply_update([self._params_with_grad[i] for i in idx], self.lr_params, 0.0, max_avg_amp, max_element_amp)
        if last:
            update_all(2 * t + 1)
        self._step += 1
        return closure_returns
