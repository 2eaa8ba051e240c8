"""One LRA preconditioner Q = (I + U V^T) diag(d) with its N rows cut over the ranks of a process group (last row).

The reference runs LRA as replicas only (psgd.py:987-1072 knows nothing of ranks).  Here U, V (N x r) and d (N) are sharded by rows: rank k
keeps rows [row0_k, row0_k + n_k).  Every stage of update_precond_lra (psgd.py:994-1052) and precond_grad_lra (:1055-1063) is row-local
except a handful of r x r / r-vector / scalar reductions over ALL rows (the Grams U^T U, V^T V, V^T U; V^T (d h), U^T (v / d); a^T U, b^T U,
a^T V, b^T V, |a|^2, |b|^2; two maxima; for the apply V^T (d g), U^T y and the sum of squares of the result).  The engine runs a stage as
PHASES (include/psgdk.h, "row shards of ONE LRA preconditioner") and leaves each phase's partial reductions in named words of its scratch
block; between two phases this module gathers those words from every rank in ONE collective, reduces them IN RANK ORDER (so every rank
forms the same bits) and writes the totals back: 4 collectives per update, 3 per apply, each of at most 3 r'^2 floats (r' = 16, 32 or 64).
The small r x r solves that consume them run replicated on every rank from identical inputs.

Unmeasured on multi-GPU hardware: tests run it over gloo (CPU stand-in engine: tests/oracle_lra_engine.py; the HIP engine with two ranks
on one GPU: tests/test_gpu_lra_sharded.py).  What the collectives cost on xGMI is priced in DESIGN.md section 6.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_rows(N: int, world: int, rank: int, align: int = 256) -> Tuple[int, int]:
    """(row0, rows) of rank's block: equal blocks of a multiple of `align` rows, the last one takes what is left (possibly nothing)."""
    per = -(-N // world)
    per = -(-per // align) * align
    row0 = min(N, rank * per)
    return row0, max(0, min(N, row0 + per) - row0)


class RowShardedLRA:
    """Drives one row shard's phase engine (lra._LraEngine after set_row_shard, or a test stand-in with the same five members:
    UPDATE_PHASES, APPLY_PHASES, scratch, segments(kind, phase), update_phase(...), apply_phase(...)) over a process group."""

    def __init__(self, engine, group=None):
        self.engine = engine
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.collectives = 0          # issued so far (tests: 4 per update, 3 per apply)

    # one collective: every rank's partial words, reduced in rank order, written back
    def _exchange(self, kind: int, phase: int):
        segs: Sequence[Tuple[int, int, int]] = self.engine.segments(kind, phase)
        if not segs or self.world == 1:
            return
        sc = self.engine.scratch
        mine = torch.cat([sc[o:o + c] for o, c, _ in segs])
        gathered = self._all_gather(mine)
        self.collectives += 1
        pos = 0
        for o, c, op in segs:
            tot = gathered[0][pos:pos + c].clone()
            for k in range(1, self.world):
                part = gathered[k][pos:pos + c]
                tot = torch.maximum(tot, part) if op == 1 else tot + part
            sc[o:o + c].copy_(tot)
            pos += c

    def _all_gather(self, mine: torch.Tensor) -> List[torch.Tensor]:
        n = mine.numel()
        if mine.is_cuda and _device_backend(self.group, mine) == "nccl":      # RCCL: in place on the device, stream-ordered
            buf = torch.empty(self.world * n, dtype=mine.dtype, device=mine.device)
            dist.all_gather_into_tensor(buf, mine, group=self.group)
            return [buf[k * n:(k + 1) * n] for k in range(self.world)]
        host = mine.detach().cpu()                                            # gloo (tests): a few hundred floats through the host
        parts = [torch.empty_like(host) for _ in range(self.world)]
        dist.all_gather(parts, host, group=self.group)
        return [p.to(mine.device) for p in parts]

    @staticmethod
    def _sum_over_ranks(x: torch.Tensor, world: int, group=None) -> torch.Tensor:
        """sum of a small tensor over the ranks, in rank order (same bits everywhere)"""
        if world == 1:
            return x
        if x.is_cuda and _device_backend(group, x) == "nccl":
            buf = torch.empty(world * x.numel(), dtype=x.dtype, device=x.device)
            dist.all_gather_into_tensor(buf, x.reshape(-1).contiguous(), group=group)
            parts = [buf[k * x.numel():(k + 1) * x.numel()].reshape(x.shape) for k in range(world)]
        else:
            host = x.detach().cpu()        # (the host copy -- a sync -- only where the transport needs it)
            parts = [torch.empty_like(host) for _ in range(world)]
            dist.all_gather(parts, host, group=group)
            parts = [p.to(x.device) for p in parts]
        tot = parts[0].clone()
        for k in range(1, world):
            tot = tot + parts[k]
        return tot

    def update_whiten(self, g_local, lr=0.1, betaL=0.9, damping=1e-9, *, v_noise=None, seed=0, offset=0, update_u=True):
        """psgd.py:1066-1072 -> 994-1052 on this rank's rows; every rank must call it with the same lr, betaL, damping, seed, offset and coin."""
        eng = self.engine
        for phase in range(eng.UPDATE_PHASES):
            eng.update_phase(phase, g_local, lr, betaL, damping, v_noise=v_noise, seed=seed, offset=offset, update_u=update_u)
            self._exchange(0, phase)

    def precond_grad(self, g_local, out: Optional[torch.Tensor] = None):
        """psgd.py:1055-1063 on this rank's rows.  Afterwards the engine's sum-of-squares word holds the sum over ALL rows."""
        eng = self.engine
        g_local = g_local.contiguous()
        if out is None:
            out = torch.empty_like(g_local)
        for phase in range(eng.APPLY_PHASES):
            eng.apply_phase(phase, g_local, out)
            self._exchange(1, phase)
        return out


def _device_backend(group, t: torch.Tensor) -> str:
    """"nccl" when RCCL will carry device tensor `t` (the same test KWNS4's exchanges use: asked of the backend FOR THE TENSOR'S DEVICE)."""
    from .kwns4 import KWNS4
    return "nccl" if KWNS4._device_backend_is_rccl(t, group) else "other"


def all_gather_rows(local: torch.Tensor, N: int, world: int, rank: int, group=None, align: int = 256) -> torch.Tensor:
    """The N-vector whose shard_rows blocks are the ranks' `local` vectors (the preconditioned gradient, gathered for the parameter update
    every rank applies in full -- the DDP surface keeps full parameters everywhere)."""
    if world == 1:
        return local
    per = -(-(-(-N // world)) // align) * align
    mine = torch.zeros(per, dtype=local.dtype, device=local.device)
    mine[:local.numel()] = local.reshape(-1)
    if local.is_cuda and _device_backend(group, local) == "nccl":
        buf = torch.empty(world * per, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(buf, mine, group=group)
    else:
        host = mine.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host, group=group)
        buf = torch.cat(parts).to(local.device)
    return buf[:N].reshape(N, 1)
