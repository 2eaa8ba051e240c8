"""The gradient reduce-scatter hook for KWNS4(shard_state=True) (psgd_torch_amd/ddp_hook.py) under a real DistributedDataParallel
model, world 2 and 3, gloo, CPU (TEST-ONLY OracleEngine for the compute): after the first iteration every bucket goes through the uneven
reduce-scatter (all_to_all_single + owner-side sum) instead of an all-reduce, both ranks end with identical parameters, and those equal
the single-process run on the concatenated batch up to the order of the gradient sums."""
import os
import socket
import sys
import tempfile

import torch
import torch.multiprocessing as mp


def _model(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(12, 16), torch.nn.Tanh(), torch.nn.Linear(16, 10), torch.nn.Tanh(), torch.nn.Linear(10, 4))


def _data(step, rank, world):
    g = torch.Generator().manual_seed(1000 + step)
    x, y = torch.randn(8 * world, 12, generator=g), torch.randn(8 * world, 4, generator=g)
    return (x, y) if rank is None else (x[rank * 8:(rank + 1) * 8], y[rank * 8:(rank + 1) * 8])


def _train(model, opt, steps, rank, world):
    for t in range(steps):
        x, y = _data(t, rank, world)
        opt.zero_grad(set_to_none=True)
        ((model(x) - y) ** 2).mean().backward()
        opt.step()


KW = dict(preconditioner_dtype=torch.float32, lr_params=1e-2, lr_preconditioner=0.3, weight_decay=0.0)


def _tall_model(seed):
    """... with a tall first layer: its (192, 12) weight has a diagonal dim-0 and a dense dim-1 factor -- the optimizer splits it by rows"""
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(12, 192), torch.nn.Tanh(), torch.nn.Linear(192, 10), torch.nn.Tanh(), torch.nn.Linear(10, 4))


def _worker(rank, world, port, outdir, tall=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        import psgd_torch_amd
        from psgd_torch_amd.ddp_hook import register_sharded_grad_hook
        from oracle_engine import OracleEngine
        model = _tall_model(3) if tall else _model(3)
        ddp = torch.nn.parallel.DistributedDataParallel(model, bucket_cap_mb=0.0005)       # several small buckets
        opt = psgd_torch_amd.KWNS4(ddp.parameters(), shard_state=True, engine_factory=OracleEngine,
                                   **dict(KW, **(dict(shard_split_rows=0.0) if tall else {})))
        st = register_sharded_grad_hook(ddp, opt)
        _train(ddp, opt, 5, rank, world)
        torch.save({"params": [p.detach().clone() for p in model.parameters()], "allreduced": st.buckets_allreduced,
                    "scattered": st.buckets_scattered, "owners": [st.owner_of(p) for p in model.parameters()]},
                   os.path.join(outdir, f"r{rank}.pt"))
    finally:
        torch.distributed.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


import pytest


@pytest.mark.parametrize("world", [2, 3])
def test_reduce_scatter_hook_matches_single_process(world):
    """(world 3: six parameters over three owners -- a bucket may hold slices of all three, or none of some rank's: empty splits)"""
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import psgd_torch_amd
    from oracle_engine import OracleEngine
    ref = _model(3)
    opt = psgd_torch_amd.KWNS4(ref.parameters(), engine_factory=OracleEngine, **KW)
    _train(ref, opt, 5, None, world)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        rs = [torch.load(os.path.join(d, f"r{r}.pt")) for r in range(world)]
    r0 = rs[0]
    assert r0["allreduced"] >= 1 and r0["scattered"] >= 4 * r0["allreduced"] > 0, (r0["allreduced"], r0["scattered"])   # first iteration only
    assert sorted(set(r0["owners"])) == list(range(world)) and all(r["owners"] == r0["owners"] for r in rs)
    for k, c in enumerate(ref.parameters()):
        for r in rs[1:]:
            assert torch.equal(r0["params"][k], r["params"][k]), "ranks diverged"
        a = r0["params"][k]
        assert torch.allclose(a, c.detach(), rtol=1e-4, atol=1e-6), float((a - c.detach()).abs().max())


def test_reduce_scatter_hook_with_a_row_split_tensor():
    """A row-split parameter (owner -1: every rank preconditions a block of its rows) makes its bucket fall back to the all-reduce --
    every rank needs its own rows of that gradient, averaged --; the other buckets are still reduce-scattered; the result equals the
    single-process run."""
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import psgd_torch_amd
    from oracle_engine import OracleEngine
    world = 2
    ref = _tall_model(3)
    opt = psgd_torch_amd.KWNS4(ref.parameters(), engine_factory=OracleEngine, **KW)
    _train(ref, opt, 5, None, world)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), d, True), nprocs=world, join=True)
        rs = [torch.load(os.path.join(d, f"r{r}.pt")) for r in range(world)]
    r0 = rs[0]
    assert r0["owners"][0] == -1 and all(o >= 0 for o in r0["owners"][1:]), r0["owners"]
    assert r0["scattered"] >= 4 and r0["allreduced"] >= 5, (r0["allreduced"], r0["scattered"])     # the split tensor's bucket on every iteration
    for k, c in enumerate(ref.parameters()):
        assert torch.equal(r0["params"][k], rs[1]["params"][k]), "ranks diverged"
        a = r0["params"][k]
        assert torch.allclose(a, c.detach(), rtol=1e-4, atol=1e-6), (k, float((a - c.detach()).abs().max()))
