"""KWNS4 -- the torch.optim.Optimizer surface of the reference's wrapped_as_torch_optimizer_for_ddp.KWNS4
(wrapped_as_torch_optimizer_for_ddp.py:4-176): same constructor arguments, defaults, assertions, param_groups /
state layout and step() semantics (no closure), driven by the batched HIP engine instead of per-tensor ATen calls.

What is different, on purpose:
  * one grouped engine call per stage for ALL parameters of a group (the reference loops over parameters in Python
    and syncs the host once per tensor at ..._ddp.py:154);
  * randomness is a private counter-based stream (CPU generator for the Bernoulli gates, Philox (seed, step) on the
    device), identical on every rank by construction -- the reference reaches the same lock-step by broadcasting and
    swapping torch RNG states (..._ddp.py:88-104,172-176);
  * `shard_state=True` (new; the reference only replicates): every rank owns a cost-balanced subset of
    the parameters, preconditions only those, and the clipped preconditioned gradients are exchanged by all-gather
    (RCCL over xGMI); every rank then applies the identical parameter update.  The tensors are worked off in
    `shard_chunks` cost-balanced chunks (default: 1 or 2 by the exchange model, `auto_shard_chunks`), each with its own exchange buffer: chunk c's asynchronous, in-place
    all-gather travels while chunk c + 1 is preconditioned, and the parameter updates follow chunk by chunk as the
    gathers land -- on a step this short the fabric, not the arithmetic, is the critical path;
  * (round 4) a DOMINANT matrix whose dim-0 factor is diagonal and whose dim-1 factor is dense (GPT-2's tied embedding) is SPLIT BY ROWS
    over all ranks (`shard_split_rows`, default on in sharded mode): every rank preconditions one row block with the replicated dense
    factor, which is fitted to the whole matrix through one small exchange in the middle of the update (the ranks' partial mode Grams,
    psgd.py:405; include/psgdk.h "row shards").
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from . import _lib as L
from .engine import KronEngine
from .sharding import assign_owners, chunk_partition, lpt_partition, kron_step_cost, row_split_candidates


_GEOMETRIES = {"Q0.5EQ1.5", "Q0p5EQ1p5", "EQ", "QEQ", "QUAD", "QEP", "QUAD4P", "PRO4P"}      # psgd.py:161 (init_kron's dQ)


def _packed(tensors):
    """The engine addresses tensors by raw pointer in logical (row-major) order; the reference's `p.subtract_(h.view_as(p))`
    (..._ddp.py:157) works for any strides.  Non-contiguous tensors (channels_last weights, transposed / tied views) are
    handed over as a packed row-major shadow; the caller copies the shadows it wrote back with `_unpack`."""
    out, back = [], []
    for t in tensors:
        if t is None or t.is_contiguous():
            out.append(t)
        else:
            sh = t.contiguous()
            out.append(sh)
            back.append((t, sh))
    return out, back


def _unpack(back):
    for t, sh in back:
        t.copy_(sh)


class _Works:
    """Several asynchronous requests behind the one-request interface of a collective's work handle."""

    def __init__(self, works, after=None):
        self.works = list(works)
        self.after = after                   # host-staged exchanges: what lands the received bytes on the device

    def wait(self):
        # idempotent: a bucket's exchange is waited for in _bucket_finish, and -- with update_preconditioner_first=False on an engine
        # with the cooperative norm bound -- once before that; a SECOND wait() on a completed gloo receive never returns
        works, self.works = self.works, []
        for w in works:
            w.wait()
        if self.after is not None:
            self.after()
            self.after = None


# Chunks of the sharded step when the caller names none.  Every chunk is a plan of its own: ~20 dependent launches whose latency floor
# (two cooperative norm bounds, stages of a handful of tiles) does not shrink with the chunk -- measured with tools/rank_arithmetic.py
# (one rank's kernels of an 8-rank job, no wire; profiles/r04_g): GPT-2-small 0.81 / 1.12 / 1.60 ms per step at 1 / 2 / 4 chunks,
# GPT-2-medium 1.91 / 2.18 / 2.94 -- about 0.3 ms per extra chunk, on the device, with the host only half as busy.  A chunk pays when the
# exchange it hides takes longer than that: at 8 ranks a GPT-2-small step receives 218 MB (~0.7 ms at 300 GB/s), so two chunks hide what
# one more plan costs, four never do (rounds 2-3 defaulted to 4 without this measurement).  bench.py's probe times 1, 2 and 4.
DEFAULT_SHARD_CHUNKS = 2


def auto_shard_chunks(numel: int, esize: int, world: int) -> int:
    """The chunk count of an optimizer that named none (round 6): ONE plan per rank unless the exchange model says a second one pays.
    A second chunk hides (at best) the first chunk's half of the exchange under its own arithmetic and costs ~0.3 ms of launches that do
    not shrink (above).  Every rank receives numel * esize / world bytes from each peer, one xGMI link per peer, all links at once:
    t = bytes per peer / link rate, priced at 77 GB/s per direction (half of the link's 153 GB/s: unmeasured on hardware, like everything
    about N > 1 here).  GPT-2-small at 8 ranks: 31 MB per peer = 0.40 ms -> one chunk; GPT-2-medium at 8: 89 MB = 1.15 ms -> two; any model
    at 2 ranks with >= 93 M elements -> two.  bench.py --parallelism auto still times 1, 2 and 4 chunks on the fabric it runs on."""
    t_exchange = numel * esize / max(world, 1) / 77e9
    return DEFAULT_SHARD_CHUNKS if 0.5 * t_exchange > 0.3e-3 else 1


class _Bucket:
    """The parameters of one param_group that share (param dtype, grad dtype): one engine."""

    def __init__(self):
        self.params: List[torch.Tensor] = []
        self.engine: Optional[KronEngine] = None
        self.owned: List[int] = []          # indices into params this rank preconditions
        self.step = 0
        self.flat = None                     # sharded mode: gathered clipped h
        self.flat_apply = None               # fused parameter update from the gathered buffer
        self.segments = None


class KWNS4(torch.optim.Optimizer):
    def __init__(
            self,
            params,
            whiten_grad=False,
            preconditioner_max_size=float("inf"),
            preconditioner_max_skew=1.0,
            preconditioner_init_scale=1.0,
            lr_params=2e-4,
            lr_preconditioner=0.5,
            betaL=0.9,
            damping=1e-9,
            momentum=0.9,
            weight_decay=0.05,
            decoupled_weight_decay=True,
            grad_clip_max_amps=(2.0, 10.0),
            preconditioner_update_probability=1.0,
            preconditioner_dtype: Optional[torch.dtype] = torch.bfloat16,
            update_preconditioner_first=True,
            resync_every=1000_000,
            *,
            dQ: str = "Q0.5EQ1.5",
            seed: int = 0,
            shard_state: bool = False,
            shard_chunks: Optional[int] = None,
            shard_exchange: str = "all_gather",
            shard_split_rows=True,
            shard_resync_every: int = 100,
            engine_factory=None,
    ):
        # the reference's argument checks, verbatim in meaning (..._ddp.py:45-62)
        assert whiten_grad in (False, True)
        assert preconditioner_max_size >= 0.0
        assert preconditioner_max_skew >= 0.0
        assert preconditioner_init_scale > 0.0
        assert lr_params > 0.0
        assert 0.0 < lr_preconditioner < 1.0
        assert 0.0 <= betaL <= 1.0
        assert damping >= 0.0
        assert 0.0 <= momentum < 1.0
        assert weight_decay >= 0.0
        assert decoupled_weight_decay in (False, True)
        assert grad_clip_max_amps[1] >= grad_clip_max_amps[0] >= 1.0
        assert 0.0 < preconditioner_update_probability <= 1.0
        assert preconditioner_dtype in (None, torch.bfloat16, torch.float32)
        assert update_preconditioner_first in (False, True)
        assert resync_every > 0
        if not whiten_grad:
            assert momentum > 0.0, "Cannot whiten momentum if momentum setting is zero."

        defaults = {
            "whiten_grad": whiten_grad,
            "preconditioner_max_size": preconditioner_max_size,
            "preconditioner_max_skew": preconditioner_max_skew,
            "preconditioner_init_scale": preconditioner_init_scale,
            "lr_params": lr_params,
            "lr_preconditioner": lr_preconditioner,
            "betaL": betaL,
            "damping": damping,
            "momentum": momentum,
            "weight_decay": weight_decay,
            "decoupled_weight_decay": decoupled_weight_decay,
            "grad_clip_max_amps": grad_clip_max_amps,
            "preconditioner_update_probability": preconditioner_update_probability,
            "preconditioner_dtype": preconditioner_dtype,
            "update_preconditioner_first": update_preconditioner_first,
            "resync_every": resync_every,
        }
        super().__init__(params, defaults)

        # the reference selects the geometry by editing three lines (..._ddp.py:84-86: dQ, update_precond, precond_grad); here it
        # is a keyword: the engine is built for it (psgd.init_kron's dQ, psgd.py:161) and dispatches update and apply on it
        if dQ not in _GEOMETRIES:
            raise NotImplementedError(f"dQ={dQ!r}: built geometries are {sorted(_GEOMETRIES)}")
        self.dQ = dQ
        self.is_distributed = torch.distributed.is_available() and torch.distributed.is_initialized()
        self.world = torch.distributed.get_world_size() if self.is_distributed else 1
        self.rank = torch.distributed.get_rank() if self.is_distributed else 0
        self.shard_state = bool(shard_state) and self.world > 1
        # sharded mode: the tensors of a bucket are worked off in this many cost-balanced chunks, each with its own exchange
        # buffer, so that chunk c's all-gather travels while chunk c + 1 is preconditioned (1 = one exchange)
        # (None: decided by auto_shard_chunks when the first bucket is built -- 0 until then)
        self._shard_chunks = (max(1, int(shard_chunks)) if shard_chunks is not None else 0) if self.shard_state else 1
        self._shard_chunks_explicit = shard_chunks is not None      # (a checkpoint may bring its own count to an optimizer that named none)
        # how a chunk's clipped preconditioned gradients travel: "all_gather" (one collective; RCCL picks the algorithm) or "p2p"
        # (every rank sends its segment to each peer directly and receives theirs: 2 (N - 1) grouped point-to-point operations, all
        # seven xGMI links of a GPU busy at once -- a ring all-gather is bound by ONE link).  bench.py --parallelism auto times both.
        assert shard_exchange in ("all_gather", "p2p")
        self._shard_exchange = shard_exchange
        assert shard_resync_every > 0
        self._shard_resync_every = int(shard_resync_every)      # period of the row-split tensors' replicated-factor resync (_bucket_finish)
        self._chunks = {}            # bucket key -> {position of the parameter in its group: chunk index}
        self._seed = int(seed)
        self._gate_gen = torch.Generator().manual_seed(self._seed)      # same stream on every rank
        self._buckets: Dict[tuple, _Bucket] = {}
        self._engine_factory = engine_factory or KronEngine
        self._global_step = 0
        self._replay = None
        self._split = set()          # bucket keys that fell back to one engine per parameter
        self._split_pd = {}
        self._split_owner = {}       # sharded mode: owner rank of every split-off parameter (kept from the batched bucket)
        self._pos_cache = {}
        self._owners = {}            # chunked buckets: bucket key -> {position of the parameter in its group: owner rank}
        self._legacy_owners = False  # a checkpoint of rounds 1-3 (owners chosen chunk by chunk) was loaded: keep that rule
        # row-split tensors: (bucket key without chunk / split suffix) -> {position in the group: [(row0, row1)] * world}; decided when the
        # key is first seen (cost model, deterministic on every rank), part of the checkpoint
        # shard_split_rows: True = tensors that cost more than half a rank's fair share of their group; a float = that fraction (0.0:
        # every tensor of the right structure); False = none
        # (the geometries whose update is the default one's up to the dense factor's own step: psgdk_plan_set_row_shard takes these three)
        self._split_rows = shard_split_rows is not False and self.shard_state and self.dQ in ("Q0.5EQ1.5", "Q0p5EQ1p5", "QEQ", "QUAD")
        self._split_rows_threshold = 0.5 if shard_split_rows is True else float(shard_split_rows or 0.0)
        self._rowsplit = {}

    # what the engine sees of a parameter: the tensors themselves here; the DTensor shell (kwns4_dtensor.py) hands over
    # the local shards (wrapped_as_torch_optimizer_for_dtensor.py:123,156)
    def _data_of(self, p: torch.Tensor) -> torch.Tensor:
        return p

    def _grad_of(self, p: torch.Tensor) -> torch.Tensor:
        return p.grad

    def _has_work(self, p: torch.Tensor) -> bool:
        return p.grad is not None                                                    # ..._ddp.py:113-115

    # hook for tests that replay the reference's recorded draws
    def _uniform(self) -> float:
        return float(torch.rand([], generator=self._gate_gen))

    def _uniforms(self, n: int) -> List[float]:
        """n successive gate draws.  One generator call instead of n: `torch.rand(n, generator=g)` consumes the CPU generator
        exactly like n calls of `torch.rand([], generator=g)` (same values, same final state -- tests/test_missing_grads.py checks
        it), and 148 scalar draws were 0.4 ms of the 0.6 ms of host time per GPT-2-small step.  A replaced `_uniform` (the tests
        that replay the reference's recorded draws set it on the instance) is honoured draw by draw."""
        if "_uniform" in self.__dict__ or type(self)._uniform is not KWNS4._uniform:
            return [self._uniform() for _ in range(n)]
        return torch.rand(n, generator=self._gate_gen).tolist()

    def _update_draws(self, b, plist):
        """The host-side draws of one batched update call: the per-tensor 1% balancing gates of psgd.py:418 (drawn for
        ALL tensors on every rank so that the gate stream stays identical across ranks); device noise is Philox unless
        a test installed `_replay` to feed the reference's recorded draws."""
        if self._replay is not None:
            return self._replay(b, plist)
        u = self._uniforms(len(plist))
        return dict(noise=None, balance_mask=[u[i] < 0.01 for i in b.owned])

    def _pos(self, gi: int, group) -> Dict[int, int]:
        """id(parameter) -> its position in the group (the Philox stream id, the chunk and split keys): built once per group and
        rebuilt when the group's parameter list changes (add_param_group, a list edited in place)."""
        c = self._pos_cache.get(gi)
        params = group["params"]
        ids = tuple(map(id, params))          # (a parameter replaced in place keeps the list's identity and length: compare the ids)
        if c is None or c[0] is not params or c[1] != ids:
            c = self._pos_cache[gi] = (params, ids, {i: k for k, i in enumerate(ids)})
        return c[2]

    # --------------------------------------------------------------------------------------------------------------
    def _buckets_for(self, gi: int, group, plist: List[torch.Tensor]):
        """[(bucket, params)] covering plist.  Normally one batched bucket -- in sharded mode `shard_chunks` of them, cut at equal
        cost, so that the exchange of one chunk overlaps the arithmetic of the next (the chunk of a parameter is fixed when its
        bucket key is first seen).  If the set of parameters with gradients changes between steps (the reference simply skips
        parameters without a gradient, ..._ddp.py:113-115) a batched bucket is SPLIT once into one single-tensor engine per
        parameter -- state carried over -- and stays split: correct for any pattern of missing gradients, at the reference's own
        launch granularity for that group."""
        p0, g0 = self._data_of(plist[0]), self._grad_of(plist[0])
        key = (gi, p0.dtype, g0.dtype, p0.device)
        pos = self._pos(gi, group)
        if self.shard_state and key not in self._rowsplit:
            # which tensors are split by rows over all ranks: decided ONCE per key from the cost model (the same on every rank)
            self._rowsplit[key] = {}
            if self._split_rows and not self._legacy_owners and not getattr(self, "_rowsplit_frozen", False):
                shapes = [tuple(self._grad_of(p).squeeze().shape) for p in plist]
                costs = [kron_step_cost(s, group["preconditioner_max_size"], group["preconditioner_max_skew"]) for s in shapes]
                cand = row_split_candidates(shapes, costs, self.world, group["preconditioner_max_size"], group["preconditioner_max_skew"],
                                            threshold=self._split_rows_threshold)
                # the blocks are cut as slices of dim 0 of the RAW gradient / parameter (x[r0:r1]): a tensor whose squeezed rows are not its
                # raw dim 0 -- (1, N, M) squeezes to (N, M) -- would hand the engine the whole tensor or nothing.  Such tensors stay whole.
                cand = {i: bl for i, bl in cand.items() if self._grad_of(plist[i]).dim() >= 1 and self._grad_of(plist[i]).shape[0] == shapes[i][0]
                        and self._grad_of(plist[i]).numel() == shapes[i][0] * shapes[i][1]}
                self._rowsplit[key] = {pos[id(plist[i])]: [tuple(b) for b in blocks] for i, blocks in cand.items()}
        if self._shard_chunks == 0:      # no count named: from the exchange model, the same on every rank (all see the same parameter list)
            esize = torch.empty((), dtype=group["preconditioner_dtype"] or g0.dtype).element_size()
            self._shard_chunks = auto_shard_chunks(sum(self._grad_of(p).numel() for p in plist), esize, self.world)
        if self._shard_chunks <= 1:
            return self._buckets_for_key(gi, group, plist, key)
        ch = self._chunks.get(key)           # {position in the group: chunk}; part of the checkpoint
        if ch is None:
            shapes = [tuple(self._grad_of(p).squeeze().shape) for p in plist]
            costs = self._unit_costs(gi, group, key, plist, shapes)
            part = chunk_partition(costs, self._shard_chunks, self.world)         # deterministic: the same on every rank
            ch = self._chunks[key] = {pos[id(p)]: c for p, c in zip(plist, part)}
            if not self._legacy_owners:
                # owners over ALL chunks at once (every rank's total, not each chunk's slowest rank, bounds the arithmetic: the
                # exchanges are asynchronous); part of the checkpoint like the chunk map.  Row-split tensors: -1 (every rank a block)
                rs = self._rowsplit.get(key, {})
                own = assign_owners(costs, part, self._shard_chunks, self.world, split=[i for i, p in enumerate(plist) if pos[id(p)] in rs])
                self._owners[key] = {pos[id(p)]: r for p, r in zip(plist, own)}
        own = self._owners.get(key)
        out = []
        for c in range(self._shard_chunks):
            sub = [p for p in plist if ch.get(pos[id(p)], 0) == c]     # (a parameter first seen later joins chunk 0, which then splits)
            if sub:
                out += self._buckets_for_key(gi, group, sub, key + ("c", c), own)
        return out

    def _unit_costs(self, gi, group, key, plist, shapes):
        """Per-rank cost of every tensor: its step cost -- ONE row block's for a row-split tensor."""
        pos = self._pos(gi, group)
        rs = self._rowsplit.get(key[:4], {})
        ms, sk = group["preconditioner_max_size"], group["preconditioner_max_skew"]
        out = []
        for p, s in zip(plist, shapes):
            blocks = rs.get(pos[id(p)])
            out.append(kron_step_cost(s, ms, sk) if blocks is None else max(kron_step_cost((b[1] - b[0], s[1]), ms, sk) for b in blocks))
        return out

    def _buckets_for_key(self, gi: int, group, plist: List[torch.Tensor], key, own=None):
        """own: {position in the group: owner rank} decided for the whole (chunked) bucket, or None (owners by the per-bucket greedy)"""
        def owners_of(ps):
            if own is None or any(pos[id(p)] not in own for p in ps):
                return None
            return [own[pos[id(p)]] for p in ps]
        pos = self._pos(gi, group)
        if key in self._split:
            out = []
            for p in plist:
                sk = key + ("p", pos[id(p)])
                if sk not in self._buckets:
                    self._bucket_for(gi, group, [p], key=sk, pd=self._split_pd.get(key), owner=self._split_owner.get(sk))
                out.append((self._buckets[sk], [p]))
            return out
        b = self._buckets.get(key)
        if b is None:
            # resumed from a checkpoint and this bucket has not been rebuilt yet: rebuild it over the parameters it HAD (some of
            # them may have no gradient on this step), so that its state is restored before the comparison below may split it
            saved = (getattr(self, "_pending_restore", None) or {}).get(self._key_str(key))
            if saved is not None and "positions" in saved and saved["positions"] != [pos[id(p)] for p in plist]:
                had = [group["params"][k] for k in saved["positions"]]

                def _dt(x):
                    return None if x in (None, "None") else getattr(torch, x.split(".")[-1])
                b = self._bucket_for(gi, group, had, key=key, shapes=[tuple(self._data_of(p).squeeze().shape) for p in had],
                                     pd=_dt(saved.get("pd")), owner=owners_of(had))
        if b is not None and b.param_ids != tuple(map(id, plist)):
            self._split_bucket(gi, group, key, b, pos)
            return self._buckets_for_key(gi, group, plist, key, own)
        return [(self._bucket_for(gi, group, plist, key=key, owner=owners_of(plist)), plist)]

    def _split_bucket(self, gi, group, key, b, pos):
        self._split.add(key)
        self._split_pd[key] = b.pd
        del self._buckets[key]
        # (sharded mode: every parameter keeps its owner; the state lives -- and is carried over -- on that rank only)
        local = {i: k for k, i in enumerate(b.owned)}
        for k, p in enumerate(b.params):
            sk = key + ("p", pos[id(p)])
            self._split_owner[sk] = [b.owner[k]]
            nb = self._bucket_for(gi, group, [p], key=sk, shapes=[b.shapes[k]], pd=b.pd, owner=[b.owner[k]])
            nb.step = b.step
            self.state[p]["step"] = b.step
            if nb.engine is None:
                continue
            oldQ, oldL = b.engine.QL(local[k])
            newQ, newL = nb.engine.QL(0)
            for src, dst in zip(oldQ, newQ):
                dst.copy_(src)
            for src, dst in zip(oldL, newL):
                dst.copy_(src)
            if group["momentum"] > 0.0:
                nb.engine.ema[0].copy_(b.engine.ema[local[k]])
            nb.engine.state_changed()
        b.engine = None

    def _bucket_for(self, gi: int, group, plist: List[torch.Tensor], key=None, shapes=None, pd=None, owner=None) -> _Bucket:
        p0 = self._data_of(plist[0])
        if key is None:
            key = (gi, p0.dtype, self._grad_of(plist[0]).dtype, p0.device)
        b = self._buckets.get(key)
        if b is not None:
            return b
        b = _Bucket()
        b.params = list(plist)
        b.param_ids = tuple(map(id, plist))
        # Philox stream ids = position of the parameter in its group: the same on every rank, whatever subset of the group
        # a rank works on (sharded ownership; DTensor ranks whose local shard of some parameter is empty)
        pos = self._pos(gi, group)
        if pd is None:
            pd = group["preconditioner_dtype"] or self._grad_of(plist[0]).dtype
        if shapes is None:
            shapes = [tuple(self._grad_of(p).squeeze().shape) for p in plist]        # ..._ddp.py:124
        b.pd = pd
        # row blocks of the row-split tensors of this bucket: {index in plist: [(row0, row1)] * world}; this rank works on block `rank`
        rs = self._rowsplit.get(key[:4], {}) if self.shard_state else {}
        b.blocks = {i: rs[pos[id(p)]] for i, p in enumerate(plist) if pos[id(p)] in rs}
        if self.shard_state:
            if owner is None:
                costs = self._unit_costs(gi, group, key, plist, shapes)
                owner = assign_owners(costs, None, 1, self.world, split=list(b.blocks))
            owner = [-1 if i in b.blocks else o for i, o in enumerate(owner)]
            b.owner = owner
            b.owned = [i for i, o in enumerate(owner) if o == self.rank or o == -1]
        else:
            b.owner = [self.rank] * len(plist)
            b.owned = list(range(len(plist)))
        b.shapes = shapes
        # what this rank's engine holds of tensor i: the tensor, or its row block
        b.rows = {i: b.blocks[i][self.rank] for i in b.blocks}
        eshape = {i: ((b.rows[i][1] - b.rows[i][0], shapes[i][1]) if i in b.rows else shapes[i]) for i in b.owned}
        if b.owned:
            p4 = self.dQ in ("QUAD4P", "PRO4P")          # the factors are P itself: init_kron squares the scale (psgd.py:186-187)
            geom = {} if self.dQ in ("Q0.5EQ1.5", "Q0p5EQ1p5") else {"geometry": self.dQ}
            if b.blocks:
                geom["row_shards"] = {k: (shapes[i][0], b.rows[i][0], self.rank, self.world) for k, i in enumerate(b.owned) if i in b.rows}
            b.engine = self._engine_factory([eshape[i] for i in b.owned], p0.device, precond_dtype=pd,
                                            max_size=group["preconditioner_max_size"],
                                            max_skew=group["preconditioner_max_skew"],
                                            use_momentum=group["momentum"] > 0.0,
                                            init_scale=group["preconditioner_init_scale"] ** (2 if p4 else 1),   # ..._ddp.py:131-137
                                            tensor_ids=[(gi << 20) + pos[id(plist[i])] for i in b.owned], **geom)
            for k, i in enumerate(b.owned):
                st = self.state[plist[i]]
                st["QL"] = b.engine.QL(k)
                st["exprs"] = (b.engine, k)
                st["ema"] = b.engine.ema[k]
        for p in plist:
            self.state[p]["step"] = 0
        if self.shard_state:
            # flat exchange buffer: equal-size (padded) segment per rank, in owner order; every tensor starts at a multiple of 8
            # elements so that the fused parameter update reads it with 16-byte accesses
            # pieces of the exchange: (tensor, row0, row1, owner) -- a whole tensor from its owner, or one row block from every rank
            pieces = []
            for i, s in enumerate(shapes):
                if i in b.blocks:
                    pieces += [(i, r0, r1, r) for r, (r0, r1) in enumerate(b.blocks[i])]
                else:
                    pieces.append((i, None, None, owner[i]))

            def pnumel(pc):
                i, r0, r1, _ = pc
                return (r1 - r0) * shapes[i][1] if r0 is not None else (math.prod(shapes[i]) if len(shapes[i]) else 1)
            numels = [pnumel(pc) for pc in pieces]
            pad8 = [(n + 7) // 8 * 8 for n in numels]
            # (round 6) the RMS clip of a row-split tensor is DEFERRED to the flat apply: every rank's segment starts with one slot of 8 elements
            # per row-split tensor of this bucket, holding the fp32 sum of h^2 of the rank's own block -- it travels with the h exchange instead
            # of in an all-reduce of its own between precond_grad and the export (engines without the deferred form keep that all-reduce)
            split_ids = sorted(b.blocks)
            b.defer_clip = bool(split_ids) and hasattr(self._engine_factory.FlatApply, "set_clip_groups")
            head = 8 * len(split_ids) if b.defer_clip else 0
            per_rank = [head + sum(pad8[k] for k, pc in enumerate(pieces) if pc[3] == r) for r in range(self.world)]
            seg = max(per_rank + [8])
            b.seg = seg
            # what each rank really has to send (elements at the head of its segment).  A dominant tensor (GPT-2's wte: 38.6 M of the
            # 124 M elements, owned by ONE rank) makes its chunk's equal-size segments mostly padding: at 8 ranks the padded gathers
            # of a GPT-2-small step deliver 714 MB to every rank where 225 MB are preconditioned gradients
            b.used = per_rank
            b.uneven = seg * self.world > 1.25 * sum(per_rank)
            b.flat = torch.zeros(self.world * seg, dtype=pd, device=p0.device)
            offs = [r * seg + head for r in range(self.world)]
            b.h_views, h_offsets = [None] * len(shapes), []
            for k, (i, r0, r1, r) in enumerate(pieces):
                view = b.flat[offs[r]:offs[r] + numels[k]].view(shapes[i] if r0 is None else (r1 - r0, shapes[i][1]))
                if r0 is None or r == self.rank:
                    b.h_views[i] = view                  # where THIS rank's engine exports tensor i (its row block of a split one)
                h_offsets.append(offs[r])
                offs[r] += pad8[k]
            b.pieces = pieces
            # p <- p (1 - wd lr) - lr h for all pieces from the gathered buffer: one launch (engine-provided)
            b.flat_apply = self._engine_factory.FlatApply(numels, h_offsets, p0.device)
            if b.defer_clip:
                esz = b.flat.element_size()
                group_of = {i: g for g, i in enumerate(split_ids)}
                b.flat_apply.set_clip_groups([group_of.get(pc[0], -1) if pc[1] is not None else -1 for pc in pieces],
                                             [8 * g * esz for g in range(len(split_ids))], seg * esz, self.world,
                                             [shapes[i][0] * shapes[i][1] for i in split_ids])
                # where this rank's engine puts its own partial sums: slot g of its segment, read as one fp32 word
                own = b.flat[self.rank * seg:self.rank * seg + head].view(torch.uint8)
                b.sum_slots = {i: own[8 * g * esz:8 * g * esz + 4].view(torch.float32) for g, i in enumerate(split_ids)}
        self._buckets[key] = b
        pending = getattr(self, "_pending_restore", None)
        if pending and self._key_str(key) in pending:      # load_state_dict() was called before this bucket existed
            self._restore_bucket(b, pending.pop(self._key_str(key)))
        return b

    @torch.no_grad()
    def step(self):
        for gi, group in enumerate(self.param_groups):
            momentum = group["momentum"]
            max_avg_amp, max_element_amp = group["grad_clip_max_amps"]
            if self._uniform() < group["preconditioner_update_probability"]:          # ..._ddp.py:109-110
                updateP_first, updateP_last = group["update_preconditioner_first"], not group["update_preconditioner_first"]
            else:
                updateP_first, updateP_last = False, False
            with_grad = [p for p in group["params"] if self._has_work(p)]             # ..._ddp.py:113-115
            if not with_grad:
                continue
            by_dtype: Dict[tuple, List[torch.Tensor]] = {}
            for p in with_grad:
                lp, lg = self._data_of(p), self._grad_of(p)
                by_dtype.setdefault((lp.dtype, lg.dtype, lp.device), []).append(p)
            # two passes over the buckets: first all the arithmetic (sharded: each bucket's exchange is started as soon as its
            # preconditioned gradients are exported, and travels while the next bucket is worked on), then the parameter updates
            items = []
            for plist in by_dtype.values():
                for b, sub in self._buckets_for(gi, group, plist):
                    items.append([b, sub, self._bucket_compute(b, group, sub, updateP_first, updateP_last, momentum, max_avg_amp,
                                                               max_element_amp), None])
            # _bucket_compute is a generator: a bucket with row-split tensors yields once, right after it has posted the exchange of
            # its partial mode Grams in the middle of its preconditioner update -- those buckets go FIRST up to that point, every other
            # bucket then runs while the records travel, and the row-split buckets finish last (their update needs the gathered
            # records).  Buckets without row-split tensors never yield.
            def run(it):
                try:
                    next(it[2])
                    return False
                except StopIteration as e:
                    it[3] = e.value
                    return True
            paused = [it for it in items if it[0].blocks and not run(it)]
            for it in items:
                if not it[0].blocks:
                    run(it)
            for it in paused:
                while not run(it):
                    pass
            # finish in the order the exchanges were POSTED (row-split buckets post theirs last): waiting for the last-posted collective
            # first would hold back the parameter updates of every chunk whose exchange landed long ago
            done_first = [it for it in items if not any(it is q for q in paused)]
            for b, sub, _, work in done_first + paused:
                self._bucket_finish(b, group, sub, work)
        self._global_step += 1
        left = getattr(self, "_pending_restore", None)
        if left and not getattr(self, "_restore_warned", True):
            # every bucket that exists by now has taken its entry; what is left belongs to buckets this step did not build
            # (parameters without a gradient so far) -- or to a checkpoint that does not match this optimizer
            import warnings
            self._restore_warned = True
            warnings.warn(f"psgd_torch_amd.KWNS4.load_state_dict: {len(left)} checkpoint bucket(s) have not been matched after the first "
                          f"step ({sorted(left)[:4]} ...): their preconditioner state is NOT restored yet (parameters without gradients "
                          "so far, or a checkpoint taken from a different parameter list / sharding)", RuntimeWarning, stacklevel=2)

    def _bucket_compute(self, b, group, plist, updateP_first, updateP_last, momentum, max_avg_amp, max_element_amp):
        """A GENERATOR (see step()): runs the bucket's arithmetic and returns the handle of its exchange (sharded mode) through
        StopIteration; yields once in the middle of the update of a bucket with row-split tensors."""
        wd, lr = group["weight_decay"], group["lr_params"]
        decoupled = group["decoupled_weight_decay"]
        t = b.step
        beta = min(t / (t + 1), momentum) if momentum > 0.0 else 0.0                  # ..._ddp.py:139-142
        src_w = L.SRC_GRAD if group["whiten_grad"] else L.SRC_EMA                     # ..._ddp.py:145
        src_p = L.SRC_GRAD if momentum == 0.0 else L.SRC_EMA                          # ..._ddp.py:150
        eng = b.engine

        def mine(i, x):      # what this rank's engine sees of tensor i: the tensor, or its row block (rows are contiguous)
            return x if i not in b.rows else x[b.rows[i][0]:b.rows[i][1]]
        shard_k = [k for k, i in enumerate(b.owned) if i in b.rows]      # engine indices of the row blocks

        def update(offset):
            """The preconditioner update of this bucket's engine; with row blocks it is cut in two around the exchange of the ranks'
            partial statistics (include/psgdk.h, "row shards"), and the generator yields while that exchange travels."""
            draws = self._update_draws(b, plist)
            if not shard_k:
                eng.update_precond(src_w, group["lr_preconditioner"], group["betaL"], group["damping"], seed=self._seed, offset=offset, **draws)
                return
            noise = draws.get("noise")
            if noise is not None:      # replayed draws come per TENSOR: this rank's rows of the damping noise
                g_noise, spd, skh = noise
                noise = ([mine(b.owned[k], g) if k in shard_k else g for k, g in enumerate(g_noise)], spd, skh)
            eng.update_begin(src_w, group["lr_preconditioner"], group["betaL"], group["damping"], seed=self._seed, offset=offset, noise=noise)
            xw = self._exchange_records(eng)
            yield
            xw.wait()
            mask = draws.get("balance_mask")
            eng.update_finish(src_w, group["lr_preconditioner"], group["betaL"], group["damping"], seed=self._seed, offset=offset, noise=noise,
                              balance_mask=mask)
            if mask is not None:      # balancing a row-split tensor needs max |q| over ALL its rows (psgd.py:266-275)
                eng.balance_shards([k for k in shard_k if mask[k]], self._reduce_max)

        if eng is not None:
            coupled = wd if (wd > 0.0 and not decoupled) else 0.0
            # the parameters themselves are needed here only for coupled weight decay and for the non-sharded update (the sharded path
            # updates ALL parameters from the gathered buffer in _bucket_finish): no packed shadows of strided parameters otherwise
            own_p, back = _packed([mine(i, self._data_of(plist[i])) for i in b.owned]) if (coupled or not self.shard_state) else (None, [])
            grads = [mine(i, self._grad_of(plist[i])) for i in b.owned]
            grads = [g if g.is_contiguous() else g.contiguous() for g in grads]
            damp = None
            if (updateP_first or updateP_last) and self._replay is None:
                # the update that follows in this step reads G + (damping + eps|G|) * noise: let the momentum pass write it
                damp = dict(source=src_w, damping=group["damping"], seed=self._seed, offset=2 * t + (0 if updateP_first else 1))
            eng.accumulate(grads, params=own_p if coupled else None, coupled_wd=coupled, beta=beta,
                           keep_grad=bool(group["whiten_grad"]) or momentum == 0.0, damp=damp)
            if updateP_first:
                yield from update(2 * t)
            fused = not self.shard_state and not shard_k and getattr(self, "_fuse_update", True) and hasattr(eng, "precond_grad_apply")
            if fused:
                # ..._ddp.py:150-157 in one call: the update of p rides in the epilogue of the product that forms h
                eng.precond_grad_apply(src_p, own_p, lr, wd if (wd > 0.0 and decoupled) else 0.0, max_avg_amp, max_element_amp)
                _unpack(back)
            else:
                eng.precond_grad(src_p)
            defer = bool(shard_k) and getattr(b, "defer_clip", False)
            if shard_k and defer:
                # the RMS clip of a row-split tensor averages over ALL its rows (..._ddp.py:153-155): this rank's sums of h^2 go into its segment
                # of the exchange buffer and the flat apply clips the blocks after the gather -- no collective of its own
                for k in shard_k:
                    b.sum_slots[b.owned[k]].copy_(eng.hsumsq[k:k + 1])
            elif shard_k:
                self._reduce_sum(eng.hsumsq, shard_k)
            if fused:
                pass
            elif not self.shard_state:
                eng.apply_update(own_p, lr, wd if (wd > 0.0 and decoupled) else 0.0, max_avg_amp, max_element_amp)
                _unpack(back)
            else:
                # all owned tensors' clipped h straight into this rank's segment of the exchange buffer: one launch
                eng.export_precond_grad([b.h_views[i] for i in b.owned], clip=2 if defer else True, max_avg_amp=max_avg_amp,
                                        max_elem_amp=max_element_amp)
        elif updateP_first or updateP_last:
            self._uniforms(len(plist))                    # keep the gate stream in lock-step with the owning ranks
        work = None
        if self.shard_state:
            work = self._exchange(b)
        if eng is not None and updateP_last:
            if work is not None and getattr(eng, "info", None) is not None and eng.info().get("nlb_coop"):
                # the cooperative norm-bound launch needs all its workgroups resident at once (one per CU); collective kernels that
                # hold CUs while it starts could run it into its spin limit (one skipped update + a permanent fall-back to the
                # multi-launch route): let the exchange land first.  The multi-launch route overlaps freely.
                work.wait()
            yield from update(2 * t + 1)
        return work

    def _exchange_records(self, eng):
        """All-gather (in place; gloo: through a clone of this rank's record) of the row blocks' exchange records: the partial mode Grams
        and diagonal maxima of include/psgdk.h "row shards".  Asynchronous."""
        x, n = eng.xchg, eng.xchg_record_bytes
        mine = x[self.rank * n:(self.rank + 1) * n]
        in_place = self._device_backend_is_rccl(x)
        return torch.distributed.all_gather_into_tensor(x, mine if in_place else mine.clone(), async_op=True)

    @staticmethod
    def _device_backend_is_rccl(t, group=None) -> bool:
        """Is the transport that will carry device tensor `t` RCCL ("nccl")?  Asked of the group's (default: WORLD's) backend FOR THE TENSOR'S DEVICE:
        init_process_group() without a backend, or with "cpu:gloo,cuda:nccl", reports a composite / undefined string through
        get_backend(), and a string compare against "nccl" would then send a real RCCL job down the host-staged path (a .cpu() sync per
        chunk) without a word."""
        try:
            pg = group if group is not None else torch.distributed.group.WORLD
            be = pg._get_backend(t.device) if hasattr(pg, "_get_backend") else None
            name = be.name() if be is not None and hasattr(be, "name") else None
            if name and name.lower() in ("nccl", "rccl"):
                return True
            if name and name.lower() in ("gloo", "mpi", "ucc"):
                return False
            # (anything else -- torch's fake group, a wrapper -- : the configured string decides)
        except Exception:      # noqa: BLE001  (fake / wrapped groups: fall back to the configured string)
            pass
        cfg = str(torch.distributed.get_backend(group)).lower()
        if t.is_cuda and "cuda:nccl" in cfg:
            return True
        return cfg == "nccl"

    def _reduce_max(self, t):
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)

    def _reduce_sum(self, hsumsq, ks):
        """Sum over the ranks of the entries `ks` of the engine's per-tensor sums of h^2 (one float each; contiguous runs in one call)."""
        for k in ks:
            torch.distributed.all_reduce(hsumsq[k:k + 1], op=torch.distributed.ReduceOp.SUM)

    def _exchange(self, b):
        """ONE exchange step of a sharded bucket: all-gather of the clipped preconditioned gradients (bf16 when the preconditioner
        is), asynchronous, so that the next chunk's arithmetic (or an update_preconditioner_first=False update) runs while the
        fabric works.  The send buffer IS this rank's segment of the receive buffer -- RCCL's in-place form: no staging copy.
        The aliasing contract of that form is checked on EVERY backend (the gloo tests run this very code): the segment must sit
        at rank * seg elements inside the gathered buffer, with the buffer's dtype and contiguous.  gloo itself does not accept
        aliased arguments, so there the (checked) segment is cloned; nothing else differs between the two transports."""
        flat, seg = b.flat, b.seg
        mine = flat[self.rank * seg:(self.rank + 1) * seg]
        assert flat.is_contiguous() and mine.is_contiguous() and flat.numel() == self.world * seg
        assert mine.data_ptr() == flat.data_ptr() + self.rank * seg * flat.element_size() and mine.dtype == flat.dtype
        if self._shard_exchange == "p2p" or b.uneven:
            # exact sizes: every rank sends the used head of its segment to each peer and receives the used heads of theirs (grouped
            # point-to-point operations).  Also taken in "all_gather" mode for a chunk whose segments
            # are more than a quarter padding -- a collective over equal-size segments would move the padding too.
            used = b.used
            ops = []
            if flat.is_cuda and not self._device_backend_is_rccl(flat):
                # Point-to-point operations are stream-ordered ONLY on RCCL (the collective stream waits for the current stream
                # when the operation is posted, and work.wait() makes the current stream wait for it).  ProcessGroupGloo's
                # send / recv take the RAW pointer: on device memory they read the segment through the BAR whenever the socket
                # is ready -- possibly before psgdk_export_precond_grad has written it (round 3's "ranks diverged") -- and
                # complete without the device knowing.  So on such a transport the exchange is HOST-STAGED: the used head of the
                # segment is copied out behind the export on the current stream (the copy returns when the bytes are on the
                # host), the requests carry host tensors, and wait() copies what arrived into the peers' segments on the current
                # stream, ahead of the parameter update that reads them.
                host_out = mine[:used[self.rank]].cpu() if used[self.rank] > 0 else None
                host_in = {}
                for r in range(self.world):
                    if r == self.rank:
                        continue
                    if host_out is not None:
                        ops.append(torch.distributed.P2POp(torch.distributed.isend, host_out, r))
                    if used[r] > 0:
                        host_in[r] = torch.empty(used[r], dtype=flat.dtype)
                        ops.append(torch.distributed.P2POp(torch.distributed.irecv, host_in[r], r))

                def land(flat=flat, seg=seg, host_in=host_in, keep=host_out):
                    for r, h in host_in.items():
                        flat[r * seg:r * seg + h.numel()].copy_(h)
                return _Works(torch.distributed.batch_isend_irecv(ops) if ops else [], after=land)
            for r in range(self.world):
                if r == self.rank:
                    continue
                if used[self.rank] > 0:
                    ops.append(torch.distributed.P2POp(torch.distributed.isend, mine[:used[self.rank]], r))
                if used[r] > 0:
                    ops.append(torch.distributed.P2POp(torch.distributed.irecv, flat[r * seg:r * seg + used[r]], r))
            return _Works(torch.distributed.batch_isend_irecv(ops) if ops else [])
        in_place = self._device_backend_is_rccl(flat)
        return torch.distributed.all_gather_into_tensor(flat, mine if in_place else mine.clone(), async_op=True)

    def _bucket_finish(self, b, group, plist, work):
        wd, lr = group["weight_decay"], group["lr_params"]
        decoupled = group["decoupled_weight_decay"]
        if self.shard_state:
            work.wait()
            present = {id(p) for p in plist}
            whole, back = _packed([self._data_of(p) if id(p) in present else None for p in b.params])
            # one entry per piece of the exchange: the tensor, or its row block (contiguous rows of the packed tensor)
            lps = [None if whole[i] is None else (whole[i] if r0 is None else whole[i][r0:r1]) for i, r0, r1, _ in b.pieces]
            if getattr(b, "defer_clip", False):
                b.flat_apply.apply(lps, b.flat, lr, wd if (wd > 0.0 and decoupled) else 0.0, clip=tuple(group["grad_clip_max_amps"]))
            else:
                b.flat_apply.apply(lps, b.flat, lr, wd if (wd > 0.0 and decoupled) else 0.0)     # ..._ddp.py:120,157
            _unpack(back)
        b.step += 1
        for p in plist:
            self.state[p]["step"] += 1
        # ..._ddp.py:163-170: periodic resync of replicated state from rank 0 (drift from non-deterministic atomics)
        if self.is_distributed and not self.shard_state and (b.step % group["resync_every"] == 0):
            self._resync(b, plist)
        # sharded mode: a row-split tensor's DENSE factor (and its L) is replicated on every member, and the members' copies drift like the
        # reference's replicas do (norm-bound sums with unordered fp32 atomics for d > 512; a cooperative norm-bound time-out on ONE member
        # makes that member skip an update its peers apply).  Same cure as ..._ddp.py:163-170, on a shorter period of its own: member 0's
        # copy replaces the others'.
        if self.shard_state and getattr(b, "blocks", None) and b.engine is not None and (b.step % self._shard_resync_every == 0):
            self._resync_row_split(b)

    def _resync_row_split(self, b):
        """Broadcast of the replicated factors of this bucket's row-split tensors from member 0: every factor of a row block that is not the
        row-sharded diagonal one (for the [diag, dense] structure row_split_candidates admits: the dense factor of dim 1) and its L."""
        eng = b.engine
        changed = False
        for k, i in enumerate(b.owned):
            if i not in b.rows:
                continue
            qs, ls = eng.QL(k)
            for f in range(1, len(qs)):                 # factor 0 is the row-sharded one: each member owns its rows of it
                q = qs[f]
                t = q.contiguous()                      # (the dense factor is a strided view of the arena: row stride = padded d)
                torch.distributed.broadcast(t, src=0)
                if t.data_ptr() != q.data_ptr():
                    q.copy_(t)
                torch.distributed.broadcast(ls[f], src=0)
                changed = True
        if changed:
            eng.state_changed()      # Q^T and the cached P = Q^T Q belong to the pre-resync factor

    def _resync(self, b, plist):
        # replicated mode: every rank holds the same bucket over the same tensors, so the whole state arena (Q, Q^T, diagonal
        # factors, L, ema of all parameters) travels as ONE broadcast instead of the reference's per-tensor ones
        for p in plist:
            torch.distributed.broadcast(p, src=0)
        if b.engine is not None:
            torch.distributed.broadcast(b.engine.state_arena, src=0)
            b.engine.state_changed()      # the cached P = Q^T Q (work arena) belongs to the pre-resync factors

    # ------------------------------------------------------------------------------------------------------------------
    # checkpoint / resume.  The reference offers none that works: its state holds opt_einsum expression objects and its
    # private RNG states live outside `state`.  Here the whole engine state of a bucket is one arena
    # tensor (Q, Qt, diagonal factors, L, ema) plus two counters and the host gate generator's state.
    def _single_device(self):
        """The one device all parameters live on, or None when they span several (then device indices stay in the checkpoint)."""
        n = sum(len(g["params"]) for g in self.param_groups)
        c = getattr(self, "_one_dev_cache", None)
        if c is None or c[0] != n:
            devs = {self._data_of(p).device for g in self.param_groups for p in g["params"]}
            c = self._one_dev_cache = (n, next(iter(devs)) if len(devs) == 1 else None)
        return c[1]

    def _dev_str(self, dev) -> str:
        """Device as it is written into a checkpoint: the bare TYPE ("cuda") when every parameter of this optimizer lives on one
        device -- the usual DDP flow is "rank 0 saves, every rank loads", and the bucket of rank r lives on cuda:r --, the full
        name otherwise."""
        return dev.type if self._single_device() is not None else str(dev)

    def _dev_from_str(self, s) -> torch.device:
        one = self._single_device()
        return one if one is not None else torch.device(s)      # (also reads version-2 files that hold "cuda:0")

    def _key_str(self, key) -> str:
        """Bucket key (group index, param dtype, grad dtype, device[, "c", chunk][, "p", position]) as a string that survives
        pickling and does not depend on WHICH GPU of its kind the optimizer lives on."""
        return "|".join([str(key[0]), str(key[1]), str(key[2]), self._dev_str(key[3])] + [str(x) for x in key[4:]])

    def state_dict(self):
        buckets = {}
        for key, b in self._buckets.items():
            gpos = {id(p): k for k, p in enumerate(self.param_groups[key[0]]["params"])}
            buckets[self._key_str(key)] = {
                "group": key[0], "n_params": len(b.params), "owned": list(b.owned), "step": b.step,
                "positions": [gpos[id(p)] for p in b.params], "pd": str(b.pd),
                "arena": b.engine.state_arena.detach().clone().cpu() if b.engine is not None else None,
            }
        # param_groups as torch.optim.Optimizer.state_dict() lays them out: hyper-parameters + 'params' as running indices
        start, groups = 0, []
        for g in self.param_groups:
            d = {k: v for k, v in g.items() if k != "params"}
            d["params"] = list(range(start, start + len(g["params"])))
            start += len(g["params"])
            groups.append(d)
        def parts(k):      # (group, param dtype, grad dtype, device[, "c", chunk][, "p", position]) in a form that survives pickling
            return [k[0], str(k[1]), str(k[2]), self._dev_str(k[3])] + list(k[4:])
        split = [{"key": self._key_str(k), "parts": parts(k), "pd": str(self._split_pd.get(k))} for k in self._split]
        split_owner = [{"parts": parts(k), "owner": list(v)} for k, v in self._split_owner.items()]
        return {"psgdk_version": 2, "state": {},      # per-parameter state lives in the bucket arenas below
                "param_groups": groups, "split": split, "split_owner": split_owner, "shard_chunks": self._shard_chunks,
                "chunks": [{"parts": parts(k), "of": dict(v)} for k, v in self._chunks.items()],
                "owners": None if self._legacy_owners else [{"parts": parts(k), "of": dict(v)} for k, v in self._owners.items()],
                "rowsplit": [{"parts": parts(k), "of": {int(q): [list(x) for x in bl] for q, bl in v.items()}} for k, v in self._rowsplit.items()],
                "dQ": self.dQ, "global_step": self._global_step, "gate_rng": self._gate_gen.get_state(), "seed": self._seed, "buckets": buckets}

    def load_state_dict(self, sd):
        """Restores a state_dict() taken from an optimizer over the SAME parameters (same order, shapes, dtypes, sharding).
        Buckets are built lazily, so this may be called right after construction; the arenas are filled when a bucket is first
        used.  Buckets that had been split into per-parameter engines (parameters without gradients on some steps) come back
        split."""
        # every check first: a checkpoint that does not fit leaves the optimizer untouched
        assert sd.get("psgdk_version") == 2, "not a psgd_torch_amd.KWNS4 state dict (version 2)"
        if sd.get("dQ", "Q0.5EQ1.5") != self.dQ:
            raise ValueError(f"checkpoint was taken with dQ={sd.get('dQ')!r}, this optimizer uses dQ={self.dQ!r}")
        adopt_chunks = None
        if sd.get("shard_chunks", 1) != self._shard_chunks:
            # the chunk count decides which tensors share a plan: the arenas of a checkpoint fit only its own count.  An optimizer that was
            # given none (the default changed from 4 to 2 in round 4) and has not built a bucket yet takes the checkpoint's
            if not (self.shard_state and not self._shard_chunks_explicit and not self._buckets):
                raise ValueError(f"checkpoint was taken with shard_chunks={sd.get('shard_chunks', 1)}, this optimizer uses {self._shard_chunks}")
            adopt_chunks = int(sd["shard_chunks"])
        assert len(sd["param_groups"]) == len(self.param_groups), "checkpoint does not match this optimizer"
        for g, saved in zip(self.param_groups, sd["param_groups"]):
            assert len(saved["params"]) == len(g["params"]), "checkpoint does not match this optimizer"
        # buckets that exist already must fit what the checkpoint holds for them (with shard_state=True the arenas and the lists of owned
        # tensors are those of ONE rank: a checkpoint is restored by the rank that wrote it)
        def _norm0(ks):
            f = ks.split("|")
            f[3] = self._dev_str(self._dev_from_str(f[3]))
            return "|".join(f)
        saved_b = {_norm0(k): v for k, v in sd["buckets"].items()}
        for key, b in self._buckets.items():
            sv = saved_b.get(self._key_str(key))
            if sv is not None and (sv["n_params"] != len(b.params) or sv["owned"] != list(b.owned)):
                raise ValueError(f"checkpoint bucket {self._key_str(key)} holds {sv['n_params']} parameters, owned {sv['owned']}; this optimizer's "
                                 f"bucket has {len(b.params)}, owned {list(b.owned)} -- a sharded checkpoint belongs to the rank that wrote it "
                                 "(same world size, same shard_chunks / shard_split_rows)")
        if adopt_chunks is not None:
            self._shard_chunks = adopt_chunks
        for g, saved in zip(self.param_groups, sd["param_groups"]):
            g.update({k: v for k, v in saved.items() if k != "params"})
        self._global_step = sd["global_step"]
        self._seed = sd["seed"]
        self._gate_gen.set_state(sd["gate_rng"])

        def _dt(s):
            return None if s == "None" else getattr(torch, s.split(".")[-1])

        def _key(parts):
            gi, pdt, gdt, dev = parts[:4]
            return (int(gi), _dt(pdt), _dt(gdt), self._dev_from_str(dev)) + tuple(parts[4:])
        for e in sd.get("split_owner", []):
            self._split_owner[_key(e["parts"])] = list(e["owner"])
        self._chunks = {}
        for e in sd.get("chunks", []):
            self._chunks[_key(e["parts"])] = {int(k): int(v) for k, v in e["of"].items()}
        # owners of chunked buckets: as saved; a checkpoint that has none (rounds 1-3) chose them chunk by chunk -- keep doing so, the
        # arenas in it belong to those owners
        self._owners = {}
        self._legacy_owners = sd.get("owners") is None
        for e in (sd.get("owners") or []):
            self._owners[_key(e["parts"])] = {int(k): int(v) for k, v in e["of"].items()}
        # row-split tensors: as saved (a checkpoint without the entry was taken before round 4: nothing is split, and stays so)
        self._rowsplit = {_key(e["parts"]): {int(q): [tuple(x) for x in bl] for q, bl in e["of"].items()} for e in (sd.get("rowsplit") or [])}
        self._rowsplit_frozen = True
        for e in sd.get("split", []):
            key = _key(e["parts"])
            if key not in self._split:
                if key in self._buckets:      # a batched bucket already exists here: drop it, its state comes from the checkpoint
                    del self._buckets[key]
                self._split.add(key)
                self._split_pd[key] = _dt(e["pd"])
        def _norm(ks):        # "gi|pdt|gdt|device|..." with the device re-written the way this optimizer writes it
            f = ks.split("|")
            f[3] = self._dev_str(self._dev_from_str(f[3]))
            return "|".join(f)
        self._pending_restore = {_norm(k): v for k, v in sd["buckets"].items()}
        self._restore_warned = False
        for key, b in self._buckets.items():
            ks = self._key_str(key)
            if ks in self._pending_restore:
                self._restore_bucket(b, self._pending_restore.pop(ks))

    def _restore_bucket(self, b, saved):
        assert saved["n_params"] == len(b.params) and saved["owned"] == list(b.owned), "checkpoint does not match this optimizer"
        b.step = saved["step"]
        for p in b.params:
            self.state[p]["step"] = saved["step"]
        if b.engine is not None:
            b.engine.state_arena.copy_(saved["arena"].to(b.engine.state_arena.device))
            b.engine.state_changed()
