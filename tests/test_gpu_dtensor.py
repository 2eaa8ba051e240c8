"""The DTensor / FSDP2 shell (psgd_torch_amd.kwns4_dtensor.KWNS4, after wrapped_as_torch_optimizer_for_dtensor.py:99-184) on the
REAL HIP engine: two ranks sharing cuda:0 (gloo transport: RCCL refuses two ranks on one device, and the transport is not what
is under test), parameters and gradients as DTensors on a 1-D mesh -- Shard(0) (one tensor has fewer rows than ranks: an empty
local shard on rank 1, ..._dtensor.py:124-125), Replicate -- with the periodic resync inside the Replicate group
(..._dtensor.py:168-179) firing during the run.  Every rank must end with what a single-process KWNS4 on the HIP engine produces
from that rank's local shards as plain tensors (the reference preconditions each local slice independently), and the replicated
parameter must be identical on both ranks."""
import os
import socket
import sys
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

FULL = [(192, 64), (1, 80), (96,), (64, 64), (130, 40)]      # (1, 80) over 2 ranks: rank 1 holds an empty shard
PLACE = ["shard", "shard", "replicate", "replicate", "shard"]
STEPS = 5


def _full(seed):
    g = torch.Generator().manual_seed(seed)
    return [0.5 * torch.randn(s, generator=g) for s in FULL]


def _local(x, kind, rank, world):
    if kind == "replicate":
        return x
    chunks = torch.chunk(x, world, dim=0)
    return chunks[rank] if rank < len(chunks) else x[:0]


def _worker(rank, world, port, outdir, pd):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch.distributed.device_mesh import init_device_mesh
        from torch.distributed.tensor import DTensor, Replicate, Shard
        import psgd_torch_amd
        from psgd_torch_amd.kwns4_dtensor import KWNS4 as DKWNS4
        dev = "cuda:0"
        # (gloo for the mesh's own group too: torch would otherwise pair cuda tensors with an RCCL group, which refuses two
        #  ranks on one device)
        mesh = init_device_mesh("cuda", (world,), backend_override={0: "gloo"})
        pl = {"shard": [Shard(0)], "replicate": [Replicate()]}

        def dt_of(full, kind):
            loc = _local(full, kind, rank, world).contiguous().to(dev)
            return DTensor.from_local(loc, mesh, pl[kind], run_check=False, shape=torch.Size(full.shape),
                                      stride=tuple(torch.empty(full.shape).stride()))

        kw = dict(preconditioner_dtype=pd, lr_params=1e-2)
        params = [torch.nn.Parameter(dt_of(x, k)) for x, k in zip(_full(7), PLACE)]
        opt = DKWNS4(params, resync_every=2, **kw)
        # comparator: the same local shards as plain tensors through the plain shell (same positions in the group, so the
        # Philox stream ids match; the empty shard simply never gets a gradient)
        plain = [torch.nn.Parameter(_local(x, k, rank, world).contiguous().to(dev).clone()) for x, k in zip(_full(7), PLACE)]
        ref = psgd_torch_amd.KWNS4(plain, **kw)
        g = torch.Generator().manual_seed(99)
        for _ in range(STEPS):
            for p, q, s, k in zip(params, plain, FULL, PLACE):
                full_g = 0.3 * torch.randn(s, generator=g)
                p.grad = dt_of(full_g, k)
                if q.numel() > 0:
                    q.grad = _local(full_g, k, rank, world).contiguous().to(dev)
            opt.step()
            ref.step()
        torch.cuda.synchronize()
        engines = [b.engine for b in opt._buckets.values() if b.engine is not None]
        out = {"local": [p.to_local().detach().cpu() for p in params], "plain": [q.detach().cpu() for q in plain],
               "n_engine_tensors": sum(e.n for e in engines), "native": all(type(e).__name__ == "KronEngine" for e in engines)}
        torch.save(out, os.path.join(outdir, f"r{rank}.pt"))
    finally:
        torch.distributed.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("pd", [torch.float32, torch.bfloat16])
def test_dtensor_shell_on_hip_engine_two_ranks_one_gpu(pd):
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), d, pd), nprocs=world, join=True)
        got = [torch.load(os.path.join(d, f"r{r}.pt")) for r in range(world)]
    # rank 1 skipped its empty shard: one tensor fewer in its engine
    assert got[0]["native"] and got[1]["native"]
    assert got[0]["n_engine_tensors"] == len(FULL) and got[1]["n_engine_tensors"] == len(FULL) - 1, [g["n_engine_tensors"] for g in got]
    tol = 2e-5 if pd == torch.float32 else 2e-2
    for rank in range(world):
        for a, b, k in zip(got[rank]["local"], got[rank]["plain"], PLACE):
            assert a.shape == b.shape
            if a.numel() == 0:
                continue
            if k == "replicate":
                continue          # resynced from rank 0 every 2 steps: compared across ranks below
            err = float((a - b).abs().max() / (b.abs().max() + 1e-12))
            assert err <= tol, (rank, k, tuple(a.shape), err)
    for i, k in enumerate(PLACE):
        if k == "replicate":
            assert torch.equal(got[0]["local"][i], got[1]["local"][i]), "replicated parameter diverged across ranks"
            err = float((got[0]["local"][i] - got[0]["plain"][i]).abs().max() / (got[0]["plain"][i].abs().max() + 1e-12))
            assert err <= tol, ("replicate vs plain", err)
