"""psgd_torch_amd -- MI355X-native engine for the PSGD Kron/LRA preconditioner hot path of lixilinx/psgd_torch.

Public surface (mirrors the reference's names for this path):
    KWNS4                                       torch.optim.Optimizer (wrapped_as_torch_optimizer_for_ddp.py:4)
    KronWhiten                                  closure-style shell (psgd.py:516)
    init_kron, update_precond_kron_whiten_q0p5eq1p5, precond_grad_kron      functional seam (psgd.py:161,394,322)
    update_precond_kron_whiten_eq                                           triangular geometry, dQ="EQ" (psgd.py:330)
    update_precond_kron_whiten_qeq, _quad, _qep                             dQ="QEQ" / "QUAD" / "QEP" (psgd.py:367, 455, 339)
    LRAWhiten, update_precond_lra_whiten, precond_grad_lra                  LRA preconditioner (psgd.py:1075,1066,1055)
    register_sharded_grad_hook                  DDP comm hook for KWNS4(shard_state=True): gradients reduce-scattered to their owners (new)
    RowShardedLRA, LRAWhiten(shard_rows=True)   one LRA preconditioner with its rows cut over the ranks (new; lra_sharded.py)
Everything computes through libpsgdk.so (hand-written HIP for gfx950, include/psgdk.h); there is no CPU fallback.
"""
from .kron import (init_kron, precond_grad_kron, update_precond_kron_whiten_eq,  # noqa: F401
                   update_precond_kron_whiten_q0p5eq1p5, update_precond_kron_whiten_qeq, update_precond_kron_whiten_quad,
                   update_precond_kron_whiten_qep, update_precond_kron_whiten_quad4p,
                   update_precond_kron_whiten_pro4p)
from .kwns4 import KWNS4  # noqa: F401
from .engine import KronEngine  # noqa: F401
from .kron_whiten import KronWhiten  # noqa: F401
from .lra import LRAWhiten, precond_grad_lra, update_precond_lra_whiten  # noqa: F401


def __getattr__(name):
    # the DDP comm hook is resolved on first use: single-GPU users never import the DistributedDataParallel side
    if name == "register_sharded_grad_hook":
        from .ddp_hook import register_sharded_grad_hook
        return register_sharded_grad_hook
    if name == "RowShardedLRA":
        from .lra_sharded import RowShardedLRA
        return RowShardedLRA
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


__all__ = ["KWNS4", "KronWhiten", "KronEngine", "init_kron", "update_precond_kron_whiten_q0p5eq1p5", "update_precond_kron_whiten_eq",
           "update_precond_kron_whiten_qeq", "update_precond_kron_whiten_quad", "update_precond_kron_whiten_qep", "update_precond_kron_whiten_quad4p", "update_precond_kron_whiten_pro4p",
           "precond_grad_kron",
           "LRAWhiten", "update_precond_lra_whiten", "precond_grad_lra", "register_sharded_grad_hook", "RowShardedLRA"]
