"""Ownership of preconditioner state across the GPUs of one node (new, build-side design: the reference's DDP wrapper
only replicates -- SURVEY C2 / 8e).  Each parameter tensor's (Q, L, ema) is an independent unit, so the path shards
by whole tensors with ONE exchange step (an all-gather of the clipped preconditioned gradients)."""
from __future__ import annotations

import math
from typing import List, Sequence


def kron_factor_kinds(shape: Sequence[int], max_size: float, max_skew: float) -> List[bool]:
    """True = dense factor, per dim; the rule of psgd.py:208."""
    numel = math.prod(shape) if len(shape) else 1
    return [not (s <= 1 or s > max_size or s * s > max_skew * numel) for s in shape]


def kron_step_cost(shape: Sequence[int], max_size: float = float("inf"), max_skew: float = 1.0) -> float:
    """Estimated seconds of one update+apply step of one tensor on one MI355X: the SURVEY 8d FLOP model at a
    conservative MFMA rate plus the streaming bytes of the elementwise stages."""
    numel = math.prod(shape) if len(shape) else 1
    flops = 0.0
    for d, dense in zip(shape, kron_factor_kinds(shape, max_size, max_skew)):
        if dense:
            apply_ = min(4.0 * numel * d, 2.0 * d ** 3 + 2.0 * numel * d)
            flops += 2 * apply_ + 2.0 * numel * d + 6.0 * d ** 3 + 512.0 * d ** 2
    return flops / 4.0e14 + numel * 40.0 / 4.0e12 + 2.0e-6


def lpt_partition(costs: Sequence[float], world: int) -> List[int]:
    """Longest-processing-time-first greedy: returns the owner rank of every item (deterministic on every rank)."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world
    owner = [0] * len(costs)
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += costs[i]
    return owner
