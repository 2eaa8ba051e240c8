"""Ownership of preconditioner state across the GPUs of one node (new, build-side design: the reference's DDP wrapper
only replicates -- SURVEY C2 / 8e).  Each parameter tensor's (Q, L, ema) is an independent unit, so the path shards
by whole tensors with ONE exchange step (an all-gather of the clipped preconditioned gradients)."""
from __future__ import annotations

import math
from typing import List, Optional, Sequence


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


def lpt_partition(costs: Sequence[float], world: int, load: Optional[List[float]] = None) -> List[int]:
    """Longest-processing-time-first greedy: returns the owner rank of every item (deterministic on every rank).
    `load`: the ranks' loads so far (updated in place) -- items go to the rank that is least loaded INCLUDING earlier calls."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    if load is None:
        load = [0.0] * world
    owner = [0] * len(costs)
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += costs[i]
    return owner


def chunk_partition(costs: Sequence[float], n_chunks: int, world: int) -> List[int]:
    """Chunk index of every tensor for the sharded step (kwns4.py): LPT over `n_chunks` chunks of equal cost, then the chunks
    numbered by the time their slowest rank needs (LPT over the ranks inside each chunk), ascending.  A chunk's all-gather can
    only start when its slowest rank has exported, and the gathers run in launch order: the chunk that holds one dominant
    tensor (GPT-2's wte: most of its chunk's cost on one rank) goes LAST, so that the other chunks' gathers are on the wire while
    that rank works through it.  Deterministic on every rank."""
    n_chunks = max(1, min(n_chunks, len(costs)))
    part = lpt_partition(costs, n_chunks)
    crit = []
    for c in range(n_chunks):
        cc = [x for x, k in zip(costs, part) if k == c]
        owner = lpt_partition(cc, world)
        crit.append(max([sum(x for x, o in zip(cc, owner) if o == r) for r in range(world)] + [0.0]))
    order = sorted(range(n_chunks), key=lambda c: (crit[c], c))
    rank_of = {c: k for k, c in enumerate(order)}
    return [rank_of[k] for k in part]


def assign_owners(costs: Sequence[float], chunk_of: Optional[Sequence[int]], n_chunks: int, world: int,
                  split: Optional[Sequence[int]] = None) -> List[int]:
    """Owner rank of every tensor of a (chunked) bucket.  Two things bound a sharded step: every rank's TOTAL over the chunks (the
    arithmetic: the chunks' exchanges are asynchronous, a rank moves on without waiting for the gather of the chunk it has just exported)
    and, per chunk, the LARGEST SEGMENT any one rank contributes (every peer pulls that segment over one xGMI link, and a chunk's
    parameter update waits for it).  So: chunk by chunk, longest tensor first, each to the rank whose load in THIS chunk plus its total
    so far is smallest, among the ranks still below their fair share of the total.  History: rounds 1-3 ran the greedy per chunk from zero loads, which hands the largest tensor
    of EVERY chunk to rank 0 (GPT-2-small, 8 ranks: 2.53 x the mean load); round 3's single greedy over all tensors levelled the totals
    but left the chunks ragged (one rank contributing 26 MB of a chunk, its neighbour nothing).  GPT-2-small, 8 ranks, wte split by rows:
    largest total 1.03 x the mean, largest single-source segment 14.5 MB (tests/test_abi_and_host.py).
    `split`: indices of ROW-SPLIT tensors (round 4): every rank owns one row block of each (costs[i] is then ONE block's cost); they are
    marked -1 and their cost sits on every rank before the greedy deals out the rest.
    `chunk_of` None: one chunk."""
    split = set(split or ())
    if chunk_of is None:
        chunk_of, n_chunks = [0] * len(costs), 1
    n = len(costs)
    fair = (sum(costs[i] for i in range(n) if i not in split) + world * sum(costs[i] for i in split)) / world
    total = [sum(costs[i] for i in split)] * world
    out = [-1] * n
    # a chunk that holds a whole tensor above HALF a rank's fair share (wte where it is NOT split) first: that tensor must land on an empty rank,
    # which then takes nothing else (ranks at their fair share are skipped while any other is below it); then the heaviest chunks
    def chunk_key(c):
        whole = [costs[i] for i in range(n) if chunk_of[i] == c and i not in split]
        mx = max(whole + [0.0])
        return (-(mx if mx > 0.5 * fair else 0.0), -sum(costs[i] for i in range(n) if chunk_of[i] == c), c)
    chunks = sorted(range(n_chunks), key=chunk_key)
    for c in chunks:
        inch = [sum(costs[i] for i in split if chunk_of[i] == c)] * world
        for i in sorted((i for i in range(n) if chunk_of[i] == c and i not in split), key=lambda i: (-costs[i], i)):
            el = [k for k in range(world) if total[k] + costs[i] <= 1.02 * fair]
            r = min(el, key=lambda k: (inch[k] + total[k], k)) if el else min(range(world), key=lambda k: (total[k], k))
            out[i] = r
            inch[r] += costs[i]
            total[r] += costs[i]
    return out


def row_blocks(rows: int, world: int, granule: int = 64) -> List[tuple]:
    """[(row0, row1)] of the `world` row blocks of a row-split tensor: equal blocks of a multiple of `granule` rows (the engine pads every
    matrix to 64 rows: whole granules waste nothing), the last one takes what is left.  None if some block would be empty."""
    per = -(-rows // world)
    per = -(-per // granule) * granule
    blocks = [(k * per, min((k + 1) * per, rows)) for k in range(world)]
    return blocks if all(b[1] > b[0] for b in blocks) else None


def row_split_candidates(shapes: Sequence[Sequence[int]], costs: Sequence[float], world: int, max_size: float = float("inf"),
                         max_skew: float = 1.0, threshold: float = 0.5) -> dict:
    """{index: blocks} of the tensors worth splitting by rows across ALL ranks (GPT-2's tied embedding is 20 % of a step's FLOPs and
    31 % of the exchanged bytes -- owned by ONE rank it caps the scaling of the arithmetic at ~5 x and makes every peer pull 77 MB from one
    source).  A tensor qualifies if it is a matrix with a DIAGONAL factor on dim 0 and a DENSE one on dim 1 (psgd.py:208; then the rows are
    independent given the dense factor, whose mode Gram is a sum over row blocks), every row block gets the same structure from its own shape,
    and its cost exceeds `threshold` of a rank's fair share of the whole group."""
    if world < 2:
        return {}
    fair = sum(costs) / world
    out = {}
    for i, (s, c) in enumerate(zip(shapes, costs)):
        if len(s) != 2 or c <= threshold * fair:
            continue
        if kron_factor_kinds(s, max_size, max_skew) != [False, True]:
            continue
        blocks = row_blocks(s[0], world)
        if blocks is None or any(kron_factor_kinds((b[1] - b[0], s[1]), max_size, max_skew) != [False, True] for b in blocks):
            continue
        out[i] = blocks
    return out
