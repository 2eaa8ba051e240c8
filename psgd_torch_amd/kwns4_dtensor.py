"""KWNS4 for DTensor / FSDP2 training -- the shell of wrapped_as_torch_optimizer_for_dtensor.py:4-184 on the HIP engine.

Same constructor as the reference's DTensor wrapper.  Every rank preconditions ITS local shard of each DTensor gradient
independently (the reference's documented choice, ..._dtensor.py:8-9): the engine is built over the local shapes, empty
local shards are skipped (:124-125), the update is applied to `p.to_local()` (:156-157).  Replicas of a shard (Replicate
placements) draw identical noise by construction -- Philox streams keyed by (seed, step, tensor id) -- instead of the
RNG-state broadcast of :88-104; the occasional resync runs inside each Replicate sub-group (:168-179)."""
from __future__ import annotations

import torch

from . import kwns4 as _base

try:
    from torch.distributed.tensor import DTensor
    from torch.distributed.tensor.placement_types import Replicate
except Exception:  # pragma: no cover  (older torch layouts)
    from torch.distributed._tensor import DTensor, Replicate


class KWNS4(_base.KWNS4):
    def __init__(self, params, whiten_grad=False, preconditioner_max_size=float("inf"), preconditioner_max_skew=1.0,
                 preconditioner_init_scale=1.0, lr_params=2e-4, lr_preconditioner=0.5, betaL=0.9, damping=1e-9, momentum=0.9,
                 weight_decay=0.05, decoupled_weight_decay=True, grad_clip_max_amps=(2.0, 10.0),
                 preconditioner_update_probability=1.0, preconditioner_dtype=torch.bfloat16, update_preconditioner_first=True,
                 resync_every=1000_000, *, seed: int = 0, engine_factory=None):
        # state is already sharded by the DTensor placements; the per-parameter ownership sharding of the DDP shell does
        # not apply here
        super().__init__(params, whiten_grad=whiten_grad, preconditioner_max_size=preconditioner_max_size,
                         preconditioner_max_skew=preconditioner_max_skew, preconditioner_init_scale=preconditioner_init_scale,
                         lr_params=lr_params, lr_preconditioner=lr_preconditioner, betaL=betaL, damping=damping,
                         momentum=momentum, weight_decay=weight_decay, decoupled_weight_decay=decoupled_weight_decay,
                         grad_clip_max_amps=grad_clip_max_amps,
                         preconditioner_update_probability=preconditioner_update_probability,
                         preconditioner_dtype=preconditioner_dtype, update_preconditioner_first=update_preconditioner_first,
                         resync_every=resync_every, shard_state=False, seed=seed, engine_factory=engine_factory)

    def _data_of(self, p):
        return p.to_local() if isinstance(p, DTensor) else p                      # ..._dtensor.py:156

    def _grad_of(self, p):
        g = p.grad
        return g.to_local() if isinstance(g, DTensor) else g                      # ..._dtensor.py:123

    def _has_work(self, p):
        return p.grad is not None and self._grad_of(p).numel() > 0                # ..._dtensor.py:114-115, 124-125

    @staticmethod
    def _bcast(view, src, pg):
        # engine views are strided windows of the state arena (padded rows, transposed storage): broadcast a packed copy
        t = view if view.is_contiguous() else view.contiguous()
        torch.distributed.broadcast(t, src=src, group=pg)
        if t is not view:
            view.copy_(t)

    def _resync(self, b, plist):
        # ..._dtensor.py:168-179: inside every Replicate sub-group of a parameter's mesh, broadcast the parameter, its ema and
        # every (Q, L) of it from the group's first rank.  Per PARAMETER, like the reference: ranks of one mesh hold different
        # sets of non-empty local shards, so the buckets (and their arenas) of two ranks need not have the same composition --
        # only the state of a replicated parameter is the same object on every rank of its Replicate group.
        touched = False
        for p in plist:
            if not isinstance(p, DTensor):
                continue
            eng_k = self.state[p].get("exprs") if p in self.state else None
            for mesh_dim, placement in enumerate(p.placements):
                if not isinstance(placement, Replicate):
                    continue
                pg = p.device_mesh.get_group(mesh_dim)
                src = torch.distributed.get_process_group_ranks(pg)[0]
                torch.distributed.broadcast(p.to_local(), src=src, group=pg)
                if eng_k is None:
                    continue
                eng, k = eng_k
                if eng.use_momentum:
                    self._bcast(eng.ema[k], src, pg)
                for q, ell in zip(*eng.QL(k)):
                    self._bcast(q, src, pg)
                    self._bcast(ell, src, pg)
                touched = True
        if touched and b.engine is not None:
            b.engine.state_changed()      # Q^T and the cached P = Q^T Q follow the received factors
