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
        self._engine: Optional[KronEngine] = None
        self._counter_m = 0
        self._step = 0
        self._whiten_grad = whiten_grad
        if not whiten_grad:
            assert self.momentum > 0, "Cannot whiten momentum if the momentum setting is invalid."
        self._seed = int(seed)
        self._gate_gen = torch.Generator().manual_seed(self._seed)

    def _uniform(self) -> float:
        return float(torch.rand([], generator=self._gate_gen))

    def _make_engine(self, grads, scale):
        p0 = self._params_with_grad[0]
        self._engine = KronEngine([tuple(g.shape) for g in grads], p0.device, precond_dtype=grads[0].dtype,
                                  max_size=self._preconditioner_max_size, max_skew=self._preconditioner_max_skew,
                                  use_momentum=True, init_scale=float(scale) ** (2 if self._dQ in ("QUAD4P", "PRO4P") else 1),
                                  geometry=self._dQ)
        self._QLs = [self._engine.QL(k) for k in range(len(grads))]

    @torch.no_grad()
    def step(self, closure):
        with torch.enable_grad():
            closure_returns = closure()
            loss = closure_returns if isinstance(closure_returns, torch.Tensor) else closure_returns[0]
            grads = [g.squeeze().contiguous() for g in torch.autograd.grad(loss, self._params_with_grad)]   # psgd.py:594-597
        if self._engine is None:
            if self._init_scale is None:                                                                     # psgd.py:599-602
                scale = max([torch.mean((torch.abs(g)) ** 4) for g in grads])
                scale = float((scale + self.damping ** 4) ** (-1 / 8))
            else:
                scale = self._init_scale
            self._make_engine(grads, scale)
        eng = self._engine
        if self.momentum > 0:                                                                                # psgd.py:604-613
            beta = min(self._counter_m / (1 + self._counter_m), self.momentum)
            self._counter_m += 1
        else:
            if self._counter_m:
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
        if (first or last) and self._dQ != "EQ":          # (the EQ update builds its own (V, Hvp) pair)
            damp = dict(source=src_w, damping=self.damping, seed=self._seed, offset=2 * t + (0 if first else 1))
        if use_m:
            eng.accumulate(grads, beta=beta, keep_grad=(src_w == L.SRC_GRAD), damp=damp)
        else:
            # no momentum: the EMA buffer is bypassed (beta = 0 would overwrite it with the gradient, which is harmless
            # because a later switch to momentum > 0 restarts from counter 0 exactly like psgd.py:612-613)
            eng.accumulate(grads, beta=0.0, keep_grad=True, damp=damp)

        def gates():
            return [self._uniform() < 0.01 for _ in grads]
        if first:
            eng.update_precond(src_w, self.lr_preconditioner, self.betaL, self.damping, seed=self._seed, offset=2 * t,
                               balance_mask=gates())
        eng.precond_grad(src_p)
        max_avg_amp, max_element_amp = self.grad_clip_max_amps                                                # psgd.py:642-651
        eng.apply_update(self._params_with_grad, self.lr_params, 0.0, max_avg_amp, max_element_amp)
        if last:
            eng.update_precond(src_w, self.lr_preconditioner, self.betaL, self.damping, seed=self._seed, offset=2 * t + 1,
                               balance_mask=gates())
        self._step += 1
        return closure_returns
