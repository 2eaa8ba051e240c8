"""world_size-2 test of the DTensor shell (wrapped_as_torch_optimizer_for_dtensor.py) on CPU with gloo.

Parameters are DTensors on a 1-D mesh: Shard(0) (incl. a tensor with fewer rows than ranks -> an empty local shard on one
rank, ..._dtensor.py:124-125) and Replicate.  Each rank must end with exactly what a single-process KWNS4 produces when fed
that rank's local shards as plain tensors (the reference preconditions each local slice independently).  The compute
engine is the TEST-ONLY OracleEngine."""
import os
import socket
import sys
import tempfile

import torch
import torch.multiprocessing as mp

FULL = [(8, 6), (1, 5), (6,), (4, 4)]           # (1, 5) sharded over 2 ranks: rank 1 holds an empty shard
PLACE = ["shard", "shard", "replicate", "shard"]


def _full(seed):
    g = torch.Generator().manual_seed(seed)
    return [0.5 * torch.randn(s, generator=g) for s in FULL]


def _worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        from torch.distributed.device_mesh import init_device_mesh
        from torch.distributed.tensor import Replicate, Shard, distribute_tensor
        from oracle_engine import OracleEngine
        from psgd_torch_amd.kwns4_dtensor import KWNS4
        mesh = init_device_mesh("cpu", (world,))
        pl = {"shard": [Shard(0)], "replicate": [Replicate()]}
        params = [torch.nn.Parameter(distribute_tensor(x, mesh, pl[k])) for x, k in zip(_full(7), PLACE)]
        opt = KWNS4(params, preconditioner_dtype=torch.float32, engine_factory=OracleEngine, lr_params=1e-2, resync_every=2)
        g = torch.Generator().manual_seed(99)
        for _ in range(4):
            for p, s, k in zip(params, FULL, PLACE):
                p.grad = distribute_tensor(0.3 * torch.randn(s, generator=g), mesh, pl[k])
            opt.step()
        torch.save([p.to_local().detach().clone() for p in params], os.path.join(outdir, f"r{rank}.pt"))
    finally:
        torch.distributed.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _local(x, kind, rank, world):
    return x if kind == "replicate" else torch.chunk(x, world, dim=0)[rank] if rank < len(torch.chunk(x, world, dim=0)) else x[:0]


def test_dtensor_shell_matches_per_shard_single_process():
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import psgd_torch_amd
    from oracle_engine import OracleEngine
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        got = [torch.load(os.path.join(d, f"r{r}.pt")) for r in range(world)]
    for rank in range(world):
        # single-process comparator over this rank's local shards (empty ones never reach the optimizer)
        locs = [_local(x, k, rank, world) for x, k in zip(_full(7), PLACE)]
        params = [torch.nn.Parameter(x.clone()) for x in locs]
        # (all four stay in the group, so that the noise stream ids -- positions in the group -- match the DTensor run)
        opt = psgd_torch_amd.KWNS4(params, preconditioner_dtype=torch.float32, engine_factory=OracleEngine, lr_params=1e-2)
        g = torch.Generator().manual_seed(99)
        for _ in range(4):
            for p, s, k in zip(params, FULL, PLACE):
                full_g = 0.3 * torch.randn(s, generator=g)
                if p.numel() > 0:
                    p.grad = _local(full_g, k, rank, world).clone()
            opt.step()
        for a, b in zip(got[rank], params):
            assert a.shape == b.shape
            assert torch.equal(a, b.data), f"rank {rank}: DTensor shell differs from the per-shard single-process result"
    # the replicated parameter stays identical across ranks
    assert torch.equal(got[0][2], got[1][2])
