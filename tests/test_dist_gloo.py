"""world_size-2 test of the N>1 (sharded) path on CPU with the gloo backend.

The compute engine is the TEST-ONLY OracleEngine (tests/oracle_engine.py) injected through KWNS4's engine_factory,
so this exercises exactly the host logic the GPU run uses: deterministic cost-balanced ownership, per-rank engines
over owned tensors only, the single all-gather exchange of the clipped preconditioned gradients, identical parameter
update on every rank, gate streams in lock-step.  The sharded result must equal the single-process result."""
import os
import socket
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

SHAPES = [(24, 16), (16,), (16, 16), (1, 8, 1), (12, 20), (20,), (8, 8), ()]


def _make(seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(0.5 * torch.randn(s, generator=g)) for s in SHAPES]


MISSING = [set(), {2}, set(), {0, 4}, {0, 4, 7}, set()]      # parameters WITHOUT a gradient, per step (the `missing` cases)


def _run(params, steps, shard, missing=False, resume=False, **kw):
    import psgd_torch_amd
    from oracle_engine import OracleEngine
    if not shard:
        kw.pop("shard_chunks", None)
        kw.pop("shard_exchange", None)

    def make():
        return psgd_torch_amd.KWNS4(params, preconditioner_dtype=torch.float32, engine_factory=OracleEngine, shard_state=shard,
                                    lr_params=1e-2, **kw)
    opt = make()
    g = torch.Generator().manual_seed(99)
    for t in range(steps):
        if resume and shard and t == 3:       # checkpoint / resume in the middle of the sharded run (after a bucket split, if any)
            sd = opt.state_dict()
            opt = make()
            opt.load_state_dict(sd)
        for i, p in enumerate(params):
            gr = 0.3 * torch.randn(p.shape, generator=g)
            p.grad = None if (missing and i in MISSING[t % len(MISSING)]) else gr
        opt.step()
    return opt


def _worker(rank, world, port, outdir, kw):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        params = _make(7)
        opt = _run(params, 6 if kw.get("missing") else 4, True, **kw)
        owned = [len(b.owned) for b in opt._buckets.values()]
        torch.save({"params": [p.data.clone() for p in params], "owned": owned, "n_buckets": len(opt._buckets),
                    "uneven": [bool(getattr(b, "uneven", False)) for b in opt._buckets.values()]},
                   os.path.join(outdir, f"r{rank}.pt"))
    finally:
        torch.distributed.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("kw", [dict(), dict(whiten_grad=True, update_preconditioner_first=False, weight_decay=0.0),
                                dict(preconditioner_update_probability=0.5, momentum=0.5),
                                dict(missing=True), dict(missing=True, update_preconditioner_first=False, weight_decay=0.02),
                                dict(shard_chunks=1), dict(shard_chunks=3, update_preconditioner_first=False),
                                dict(missing=True, shard_chunks=2), dict(resume=True), dict(missing=True, resume=True),
                                dict(shard_exchange="p2p"), dict(shard_exchange="p2p", shard_chunks=1, update_preconditioner_first=False)])
def test_sharded_equals_replicated(kw):
    """(missing=True: some parameters have no gradient on some steps -- the reference skips them, ..._ddp.py:113-115; the
    sharded optimizer splits its bucket per parameter, every parameter keeping its owner, and skips their update and decay.)"""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    ref_params = _make(7)
    _run(ref_params, 6 if kw.get("missing") else 4, False, **{k: v for k, v in kw.items() if k != "resume"})
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, _free_port(), d, kw), nprocs=2, join=True)
        r0 = torch.load(os.path.join(d, "r0.pt"))
        r1 = torch.load(os.path.join(d, "r1.pt"))
    if not kw.get("missing"):
        assert sum(r0["owned"]) + sum(r1["owned"]) == len(SHAPES) and min(sum(r0["owned"]), sum(r1["owned"])) >= 1
        # (default: 4 chunks per bucket, each with its own exchange; shard_chunks=1: one)
        assert r0["n_buckets"] == min(kw.get("shard_chunks", 4), len(SHAPES)), r0["n_buckets"]
        # both exchange forms are exercised: one chunk over two ranks is balanced (the collective over equal segments), the small
        # chunks of the default setting are mostly padding (exact-size point-to-point exchange)
        if kw.get("shard_chunks") == 1 and kw.get("shard_exchange") != "p2p":
            assert not any(r0["uneven"]), r0["uneven"]
        if "shard_chunks" not in kw:
            assert any(r0["uneven"]), r0["uneven"]
    for a, b, c in zip(r0["params"], r1["params"], ref_params):
        assert torch.equal(a, b), "ranks diverged"
        assert torch.allclose(a, c.data, rtol=0, atol=0), "sharded result differs from the single-process result"


@pytest.mark.parametrize("kw", [dict(), dict(shard_exchange="p2p", update_preconditioner_first=False), dict(missing=True, resume=True, shard_chunks=2)])
def test_sharded_world_of_three(kw):
    """An odd world size: ownership cannot be even (8 tensors over 3 ranks, a rank may own nothing of a chunk), every rank has TWO
    peers in the point-to-point exchange, and the padded exchange segments differ per chunk.  Same bar as the world-2 test: all ranks
    bitwise equal, and equal to the single-process run."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    ref_params = _make(7)
    _run(ref_params, 6 if kw.get("missing") else 4, False, **{k: v for k, v in kw.items() if k != "resume"})
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(3, _free_port(), d, kw), nprocs=3, join=True)
        rs = [torch.load(os.path.join(d, f"r{r}.pt")) for r in range(3)]
    if not kw.get("missing"):
        assert sum(sum(r["owned"]) for r in rs) == len(SHAPES)
    for k, c in enumerate(ref_params):
        for r in rs[1:]:
            assert torch.equal(rs[0]["params"][k], r["params"][k]), "ranks diverged"
        assert torch.equal(rs[0]["params"][k], c.data), "sharded result differs from the single-process result"
