"""A DistributedDataParallel communication hook for `KWNS4(shard_state=True)`: gradients are REDUCE-SCATTERED to the owners of the
parameters instead of all-reduced to everybody.

Why.  Under plain DDP every gradient bucket is all-reduced -- a reduce-scatter followed by an all-gather, 4 bytes per
parameter each way -- and the sharded optimizer then exchanges the clipped preconditioned gradients h (2 bytes per parameter in bf16) with a
second all-gather.  But only the OWNER of a parameter reads its gradient (`KWNS4._bucket_compute` hands the engine `grads[i] for i in
b.owned`), so the all-gather half of the gradient all-reduce carries bytes nobody reads.  With this hook each bucket does only the
reduce-scatter half, to the owners; the optimizer's all-gather of h then takes the place of the dropped half at half its size: the sharded
step costs the training iteration no extra traffic at all -- a quarter LESS than plain DDP.  The reference (replicas only,
wrapped_as_torch_optimizer_for_ddp.py:88-104) has no counterpart.

How.  Owners are per parameter, buckets are flat and hold several parameters, so the reduce-scatter is uneven: the bucket is permuted
into owner order, `all_to_all_single` with per-owner split sizes delivers to every rank the N copies of the slices it owns (the same
bytes on the wire as a reduce-scatter: (N - 1) / N of the bucket out and in), and the owner sums them in rank order -- on every backend
alike, so the result does not depend on the transport's reduction order -- divides by N (DDP's averaging) and writes the result into
its parameters' slices of the bucket.  The other slices keep this rank's LOCAL gradients: nobody reads them (`zero_grad` or the next
backward overwrites them).

Ownership is decided by the optimizer at its first step (cost-balanced chunks, `sharding.py`), i.e. after the first backward; until then
the hook falls back to an ordinary all-reduce.  Usage:

    ddp = torch.nn.parallel.DistributedDataParallel(model, ...)
    opt = KWNS4(ddp.parameters(), shard_state=True, ...)
    register_sharded_grad_hook(ddp, opt)
"""
from typing import Dict

import torch
import torch.distributed as dist


class ShardedGradHookState:
    """What the hook needs between calls (also the place tests look at: how many buckets went which way)."""

    def __init__(self, optimizer, process_group=None):
        self.opt = optimizer
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        # owners are ranks of the OPTIMIZER's group (the default group): the hook's group must number the ranks the same way
        assert self.world == optimizer.world and self.rank == optimizer.rank, "the comm hook's process group must be the optimizer's"
        self.buckets_allreduced = 0
        self.buckets_scattered = 0
        self._owner: Dict[int, int] = {}
        self._owner_gen = None

    def owner_of(self, p) -> int:
        """Owner rank of parameter p; -1 = every rank needs its gradient (a row-split tensor: each rank preconditions a row block);
        -2 while the optimizer has not built the bucket that holds it.  The map is rebuilt whenever the optimizer's set of buckets
        changes (a bucket split per parameter, load_state_dict)."""
        gen = tuple(id(b) for b in self.opt._buckets.values())
        if gen != self._owner_gen:
            self._owner = {id(q): b.owner[k] for b in self.opt._buckets.values() for k, q in enumerate(b.params)}
            self._owner_gen = gen
        return self._owner.get(id(p), -2)


def sharded_grad_hook(state: ShardedGradHookState, bucket: dist.GradBucket) -> torch.futures.Future[torch.Tensor]:
    # (DDP checks these annotations at registration: real classes, not strings)
    flat = bucket.buffer()
    world = state.world
    grads = bucket.gradients()                    # views of `flat`, in the order of bucket.parameters()
    owners = [state.owner_of(p) for p in bucket.parameters()]
    if any(o < 0 for o in owners) or not state.opt.shard_state:
        # ownership not decided yet (first iteration), a row-split tensor in the bucket (every rank needs all of its gradient's rows
        # it owns -- and they are spread over all ranks), or the optimizer is not sharded: DDP's default behaviour
        state.buckets_allreduced += 1
        fut = dist.all_reduce(flat, group=state.group, async_op=True).get_future()
        return fut.then(lambda f: f.value()[0].div_(world))
    state.buckets_scattered += 1
    by_owner = [[g for g, o in zip(grads, owners) if o == r] for r in range(world)]
    in_splits = [sum(g.numel() for g in gs) for gs in by_owner]
    mine = in_splits[state.rank]
    pieces = [g.reshape(-1) for gs in by_owner for g in gs]
    send = torch.cat(pieces) if pieces else flat.new_empty(0)          # the bucket in owner order
    recv = flat.new_empty(world * mine)
    work = dist.all_to_all_single(recv, send, output_split_sizes=[mine] * world, input_split_sizes=in_splits, group=state.group,
                                  async_op=True)

    def finish(_):
        # (on RCCL this callback runs on a pool stream, not the stream `recv` and `send` were allocated on: tell the caching allocator,
        #  or it may hand the blocks out again while the sum below is still pending)
        if recv.is_cuda:
            recv.record_stream(torch.cuda.current_stream())
            send.record_stream(torch.cuda.current_stream())
        if mine:
            red = recv.view(world, mine).sum(dim=0).div_(world)           # rank order: the same sum on every transport
            off = 0
            for g in by_owner[state.rank]:
                g.copy_(red[off:off + g.numel()].view_as(g))
                off += g.numel()
        return flat
    return work.get_future().then(finish)


def register_sharded_grad_hook(ddp_model, optimizer, process_group=None) -> ShardedGradHookState:
    """Registers the hook on a DistributedDataParallel model whose parameters `optimizer` (a KWNS4 with shard_state=True) updates.
    Returns the hook's state object (counters `buckets_allreduced` / `buckets_scattered`)."""
    if not getattr(optimizer, "shard_state", False):
        raise ValueError("register_sharded_grad_hook needs KWNS4(shard_state=True) under an initialised process group with world size > 1")
    state = ShardedGradHookState(optimizer, process_group)
    ddp_model.register_comm_hook(state, sharded_grad_hook)
    return state
