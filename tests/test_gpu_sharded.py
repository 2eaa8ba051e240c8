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


ROW_SHAPES = [(320, 72), (72,), (72, 72), (1, 48, 1), (40, 72), (200, 64), (3, 4, 5), (), (96, 64)]      # (320, 72), (200, 64): split by rows


def _row_worker(rank, world, port, outdir, kw):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import psgd_torch_amd
        g = torch.Generator().manual_seed(7)
        params = [torch.nn.Parameter((0.5 * torch.randn(s, generator=g)).to("cuda:0")) for s in ROW_SHAPES]
        kw = dict(kw)
        force = kw.pop("_force_balance", False)
        opt = psgd_torch_amd.KWNS4(params, shard_state=True, shard_split_rows=0.0, lr_params=1e-2, **kw)
        if force:
            opt._update_draws = lambda b, plist: dict(noise=None, balance_mask=[True] * len(b.owned))
        g = torch.Generator().manual_seed(99)
        for _ in range(5):
            for p in params:
                p.grad = (0.3 * torch.randn(p.shape, generator=g)).to("cuda:0")
            opt.step()
        torch.cuda.synchronize()
        split = sorted(i for b in opt._buckets.values() for i, p in enumerate(params) if any(p is b.params[j] for j in b.blocks))
        # the replicated dense factor of the first split tensor, as this rank holds it
        q2 = opt.state[params[0]]["QL"][0][1].float().cpu()
        q1 = opt.state[params[0]]["QL"][0][0].float().cpu()
        torch.save({"params": [p.data.cpu() for p in params], "split": split, "q2": q2, "q1_rows": q1.numel()}, os.path.join(outdir, f"r{rank}.pt"))
    finally:
        torch.distributed.destroy_process_group()


@pytest.mark.parametrize("kw", [dict(preconditioner_dtype=torch.float32), dict(preconditioner_dtype=torch.bfloat16, whiten_grad=True, shard_chunks=1),
                                dict(preconditioner_dtype=torch.float32, update_preconditioner_first=False, _force_balance=True),
                                dict(preconditioner_dtype=torch.float32, dQ="QEQ"), dict(preconditioner_dtype=torch.bfloat16, dQ="QUAD"),
                                dict(preconditioner_dtype=torch.float32, dQ="QUAD", _force_balance=True)])
def test_row_split_hip_engine_two_ranks_one_gpu(kw):
    """Row-split tensors on the REAL engine (include/psgdk.h "row shards"): two ranks on cuda:0 over gloo, two matrices split by rows --
    phased update around the exchange of the partial mode Grams, the diagonal factor's maximum over both blocks, two-phase balancing
    (third case: every gate fires), the RMS clip from both blocks' sums of h^2.  The ranks agree bitwise on every parameter; against
    the replicated single-process optimizer: unsplit tensors as in the test above, the split ones within the same bounds (their mode
    Gram is summed in another order: fp32 partials of the two blocks instead of one K loop); the replicated dense factor is the same on
    both ranks up to the order of the norm bounds' fp32 atomics.  (Round 6: also the QEQ and QUAD geometries, whose phased update is the
    same up to the dense factor's own step.)"""
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import psgd_torch_amd
    rkw = {k: v for k, v in kw.items() if k not in ("shard_chunks", "_force_balance")}
    g = torch.Generator().manual_seed(7)
    ref = [torch.nn.Parameter((0.5 * torch.randn(s, generator=g)).to("cuda:0")) for s in ROW_SHAPES]
    opt = psgd_torch_amd.KWNS4(ref, lr_params=1e-2, **rkw)
    if kw.get("_force_balance"):
        opt._update_draws = lambda b, plist: dict(noise=None, balance_mask=[True] * len(b.owned))
    g = torch.Generator().manual_seed(99)
    for _ in range(5):
        for p in ref:
            p.grad = (0.3 * torch.randn(p.shape, generator=g)).to("cuda:0")
        opt.step()
    torch.cuda.synchronize()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_row_worker, args=(2, _free_port(), d, kw), nprocs=2, join=True)
        r0 = torch.load(os.path.join(d, "r0.pt"))
        r1 = torch.load(os.path.join(d, "r1.pt"))
    assert r0["split"] == [0, 5] and r1["split"] == [0, 5], (r0["split"], r1["split"])
    assert r0["q1_rows"] + r1["q1_rows"] == ROW_SHAPES[0][0]
    bf16 = kw["preconditioner_dtype"] == torch.bfloat16
    tol = 2e-2 if bf16 else 2e-5
    assert float((r0["q2"] - r1["q2"]).abs().max() / r0["q2"].abs().max()) <= (2e-2 if bf16 else 1e-5), "the replicated dense factor drifted"
    for k, (a, b, c) in enumerate(zip(r0["params"], r1["params"], ref)):
        assert torch.equal(a, b), ("ranks diverged", k)
        err = float((a - c.data.cpu()).abs().max() / (c.data.abs().max().cpu() + 1e-12))
        assert err <= tol, ("sharded vs replicated", k, err)
