"""The sharded (N > 1) path with the REAL HIP engine: two ranks sharing cuda:0, gloo transport (RCCL refuses two ranks on
one device; the transport is not what is under test).  Per-parameter ownership, per-rank engines over owned tensors,
export of the clipped preconditioned gradients into the flat exchange buffer, one all-gather, identical parameter update:
both ranks must agree bitwise with each other and match the single-process (replicated) optimizer."""
import os
import socket
import sys
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

SHAPES = [(96, 64), (64,), (64, 64), (1, 48, 1), (40, 72), (72,), (3, 4, 5), (), (96, 64), (64, 64), (40, 72), (48,)]


def _make(seed, dev):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter((0.5 * torch.randn(s, generator=g)).to(dev)) for s in SHAPES]


def _run(params, steps, shard, dev, **kw):
    import psgd_torch_amd
    opt = psgd_torch_amd.KWNS4(params, shard_state=shard, lr_params=1e-2, **kw)
    g = torch.Generator().manual_seed(99)
    for _ in range(steps):
        for p in params:
            p.grad = (0.3 * torch.randn(p.shape, generator=g)).to(dev)
        opt.step()
    torch.cuda.synchronize()
    return opt


def _worker(rank, world, port, outdir, kw):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        params = _make(7, "cuda:0")
        opt = _run(params, 5, True, "cuda:0", **kw)
        owned = [len(b.owned) for b in opt._buckets.values()]
        torch.save({"params": [p.data.cpu() for p in params], "owned": owned,
                    "uneven": [bool(b.uneven) for b in opt._buckets.values()]}, os.path.join(outdir, f"r{rank}.pt"))
    finally:
        torch.distributed.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("exchange", ["all_gather", "p2p"])
@pytest.mark.parametrize("kw", [dict(preconditioner_dtype=torch.float32), dict(preconditioner_dtype=torch.bfloat16, whiten_grad=True)])
def test_sharded_hip_engine_two_ranks_one_gpu(kw, exchange):
    """Both exchange modes, one chunk and four: with one chunk the two ranks' segments are level (the in-place collective in
    "all_gather" mode), with four every chunk's tensors sit on one rank (exact-size point-to-point exchange in BOTH modes --
    host-staged under gloo, see KWNS4._exchange).  The ranks must agree bitwise; both kinds of chunk must have occurred."""
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    ref = _make(7, "cuda:0")
    _run(ref, 5, False, "cuda:0", **kw)
    seen = set()
    for chunks in (1, 4):
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(_worker, args=(2, _free_port(), d, dict(kw, shard_exchange=exchange, shard_chunks=chunks)), nprocs=2, join=True)
            r0 = torch.load(os.path.join(d, "r0.pt"))
            r1 = torch.load(os.path.join(d, "r1.pt"))
        assert sum(r0["owned"]) + sum(r1["owned"]) == len(SHAPES) and min(sum(r0["owned"]), sum(r1["owned"])) >= 1
        assert r0["uneven"] == r1["uneven"] and len(r0["uneven"]) == chunks
        seen |= set(r0["uneven"])
        tol = 1e-5 if kw["preconditioner_dtype"] == torch.float32 else 2e-2
        for a, b, c in zip(r0["params"], r1["params"], ref):
            assert torch.equal(a, b), ("ranks diverged", exchange, chunks)
            err = float((a - c.data.cpu()).abs().max() / (c.data.abs().max().cpu() + 1e-12))
            assert err <= tol, ("sharded vs replicated", exchange, chunks, err)
    assert seen == {True, False}, ("both padded-away (exact-size) and level (collective) chunks must have been exchanged", seen)
