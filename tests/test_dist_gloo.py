"""world_size-2 test of the N>1 (sharded) path on CPU with the gloo backend.

The compute engine is the TEST-ONLY OracleEngine (tests/oracle_engine.py) injected through KWNS4's engine_factory,
so this exercises exactly the host logic the GPU run uses: deterministic cost-balanced ownership, per-rank engines
over owned tensors only, the single all-gather exchange of the clipped preconditioned gradients, identical parameter
update on every rank, gate streams in lock-step.  The sharded result must equal the single-process result."""
import math
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
    kw = dict(kw)
    if kw.pop("_coop", False):
        # like the HIP engine with the cooperative norm bound: KWNS4 then waits for a bucket's exchange BEFORE an update that follows it
        # (update_preconditioner_first=False) and once more in _bucket_finish -- a second wait() on a gloo receive used to hang
        class OracleEngine(OracleEngine):
            def info(self):
                return {"nlb_coop": 1}
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
            if resume == "default_chunks":    # the resuming optimizer names no chunk count: it takes the checkpoint's
                kw.pop("shard_chunks")
            opt = make()
            opt.load_state_dict(sd)
            if resume == "default_chunks":
                assert opt._shard_chunks == sd["shard_chunks"] == 4
                import pytest
                with pytest.raises(ValueError):       # ... one that names another count is refused
                    psgd_torch_amd.KWNS4(params, preconditioner_dtype=torch.float32, engine_factory=OracleEngine, shard_state=True,
                                         shard_chunks=3).load_state_dict(sd)
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


@pytest.mark.parametrize("kw", [dict(), dict(shard_chunks=4), dict(whiten_grad=True, update_preconditioner_first=False, weight_decay=0.0, shard_chunks=4),
                                dict(preconditioner_update_probability=0.5, momentum=0.5),
                                dict(missing=True), dict(missing=True, update_preconditioner_first=False, weight_decay=0.02),
                                dict(shard_chunks=1), dict(shard_chunks=3, update_preconditioner_first=False),
                                dict(missing=True, shard_chunks=2), dict(resume=True), dict(missing=True, resume=True),
                                dict(resume="default_chunks", shard_chunks=4),
                                dict(shard_exchange="p2p"), dict(shard_exchange="p2p", shard_chunks=1, update_preconditioner_first=False),
                                dict(update_preconditioner_first=False, _coop=True), dict(update_preconditioner_first=False, _coop=True, shard_exchange="p2p")])
def test_sharded_equals_replicated(kw):
    """(missing=True: some parameters have no gradient on some steps -- the reference skips them, ..._ddp.py:113-115; the
    sharded optimizer splits its bucket per parameter, every parameter keeping its owner, and skips their update and decay.)"""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    ref_params = _make(7)
    _run(ref_params, 6 if kw.get("missing") else 4, False, **{k: v for k, v in kw.items() if k not in ("resume", "_coop")})
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, _free_port(), d, kw), nprocs=2, join=True)
        r0 = torch.load(os.path.join(d, "r0.pt"))
        r1 = torch.load(os.path.join(d, "r1.pt"))
    if not kw.get("missing"):
        assert sum(r0["owned"]) + sum(r1["owned"]) == len(SHAPES) and min(sum(r0["owned"]), sum(r1["owned"])) >= 1
        # (default: 1 or 2 chunks per bucket by the exchange model -- one for a model of this size --, each with its own exchange)
        from psgd_torch_amd.kwns4 import DEFAULT_SHARD_CHUNKS, auto_shard_chunks
        assert auto_shard_chunks(sum(math.prod(s) for s in SHAPES), 4, 2) == 1
        assert auto_shard_chunks(124_475_904, 2, 8) == 1 and auto_shard_chunks(354_871_296, 2, 8) == DEFAULT_SHARD_CHUNKS == 2
        assert auto_shard_chunks(124_475_904, 2, 2) == 2
        assert r0["n_buckets"] == min(kw.get("shard_chunks", 1), len(SHAPES)), r0["n_buckets"]
        # both exchange forms are exercised: one chunk over two ranks is balanced (the collective over equal segments), the small
        # chunks of a four-chunk setting are mostly padding (exact-size point-to-point exchange)
        if kw.get("shard_chunks") == 1 and kw.get("shard_exchange") != "p2p":
            assert not any(r0["uneven"]), r0["uneven"]
        if kw.get("shard_chunks") == 4:
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


# ---- row-split tensors (round 4): a dominant matrix with a diagonal dim-0 factor and a dense dim-1 factor is split by rows over ALL ranks
ROW_SHAPES = [(320, 24), (24,), (24, 24), (1, 8, 1), (12, 20), (20,), (200, 16), ()]      # (320, 24) and (200, 16) qualify (psgd.py:208)


def _row_worker(rank, world, port, outdir, kw, shapes, steps):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        import psgd_torch_amd
        from oracle_engine import OracleEngine
        g = torch.Generator().manual_seed(7)
        params = [torch.nn.Parameter(0.5 * torch.randn(s, generator=g)) for s in shapes]
        kw = dict(kw)
        resume = kw.pop("resume", False)
        force = kw.pop("_test_force_balance", False)

        def make():
            o = psgd_torch_amd.KWNS4(params, preconditioner_dtype=torch.float32, engine_factory=OracleEngine, shard_state=True,
                                     lr_params=1e-2, shard_split_rows=0.0, **kw)
            if force:      # every balancing gate fires
                o._update_draws = lambda b, plist: dict(noise=None, balance_mask=[True] * len(b.owned))
            return o
        opt = make()
        # (round 6) the sums of h^2 of a row-split tensor's blocks ride inside the h exchange (deferred clip): the only all-reduce a step may
        # still issue is the balancing maximum, on the 1 % of steps whose gate fires
        n_allreduce = [0]
        real_all_reduce = torch.distributed.all_reduce

        def counting_all_reduce(*a, **k):
            n_allreduce[0] += 1
            return real_all_reduce(*a, **k)
        torch.distributed.all_reduce = counting_all_reduce
        g = torch.Generator().manual_seed(99)
        for t in range(steps):
            if resume and t == 2:
                sd = opt.state_dict()
                opt = make()
                opt.load_state_dict(sd)
            for p in params:
                p.grad = 0.3 * torch.randn(p.shape, generator=g)
            opt.step()
        torch.distributed.all_reduce = real_all_reduce
        assert force or n_allreduce[0] == 0, n_allreduce[0]
        split = sorted(i for b in opt._buckets.values() for i, p in enumerate(params) if any(p is b.params[j] for j in b.blocks))
        load = sum(len(b.owned) for b in opt._buckets.values())
        torch.save({"params": [p.data.clone() for p in params], "split": split, "load": load}, os.path.join(outdir, f"r{rank}.pt"))
    finally:
        torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world,kw", [(2, dict()), (2, dict(shard_chunks=1, update_preconditioner_first=False, whiten_grad=True)),
                                      (3, dict(preconditioner_update_probability=0.6, momentum=0.5)), (2, dict(resume=True)),
                                      (2, dict(_force_balance=True)), (2, dict(dQ="QEQ")), (3, dict(dQ="QUAD")),
                                      (2, dict(dQ="QUAD", _force_balance=True))])
def test_row_split_matches_single_process(world, kw):
    """Tensors split by rows over all ranks: the ranks agree BITWISE with each other; tensors that are not split equal the
    single-process result bitwise; the split ones to fp32 rounding (their dense factor's mode Gram is the sum of the ranks' partial
    Grams -- another order of the same additions -- and the RMS of h is formed from the ranks' partial sums)."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import psgd_torch_amd
    from oracle_engine import OracleEngine
    kw = dict(kw)
    force = kw.pop("_force_balance", False)
    steps = 4
    g = torch.Generator().manual_seed(7)
    ref = [torch.nn.Parameter(0.5 * torch.randn(s, generator=g)) for s in ROW_SHAPES]
    rkw = {k: v for k, v in kw.items() if k not in ("shard_chunks", "resume")}
    if force:
        # (every balancing gate fires: the two-phase balancing of the row blocks runs on every step)
        kw["_test_force_balance"] = True
    opt = psgd_torch_amd.KWNS4(ref, preconditioner_dtype=torch.float32, engine_factory=OracleEngine, lr_params=1e-2, **rkw)
    if force:
        opt._update_draws = lambda b, plist: dict(noise=None, balance_mask=[True] * len(b.owned))
    g = torch.Generator().manual_seed(99)
    for _ in range(steps):
        for p in ref:
            p.grad = 0.3 * torch.randn(p.shape, generator=g)
        opt.step()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_row_worker, args=(world, _free_port(), d, kw, ROW_SHAPES, steps), nprocs=world, join=True)
        rs = [torch.load(os.path.join(d, f"r{r}.pt")) for r in range(world)]
    from psgd_torch_amd.sharding import row_blocks
    assert rs[0]["split"] == [k for k in (0, 6) if row_blocks(ROW_SHAPES[k][0], world) is not None] and 0 in rs[0]["split"], rs[0]["split"]
    for k, c in enumerate(ref):
        for r in rs[1:]:
            assert torch.equal(rs[0]["params"][k], r["params"][k]), ("ranks diverged", k)
        if k in rs[0]["split"]:
            err = float((rs[0]["params"][k] - c.data).abs().max() / c.data.abs().max())
            assert err <= 2e-6 * steps, ("row-split tensor vs the single-process result", k, err)
        else:
            # (bitwise in the small-shape tests above; at these sizes the CPU stand-in's fp32 matmuls are not reproducible bit for bit
            #  ACROSS PROCESSES -- the BLAS picks its kernels by operand alignment -- so: one or two ulp)
            err = float((rs[0]["params"][k] - c.data).abs().max() / c.data.abs().max())
            assert err <= 3e-7 * steps, ("unsplit tensor differs from the single-process result", k, err)


def test_row_split_world_of_eight_on_a_gpt2_shaped_list():
    """Eight ranks, the parameter list of a (scaled-down) GPT-2 -- tied embedding, position embedding, two blocks of twelve tensors, final
    norm: 28 tensors in four chunks.  The embedding is split by rows over all eight ranks (every rank a block; the last block is the
    remainder), everything else has one owner.  All ranks bitwise equal; equal to the single-process run to fp32 rounding.  (The full-size owner map -- loads, exchange segments -- is checked in test_abi_and_host.py.)"""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    for q in (here, root):
        if q not in sys.path:
            sys.path.insert(0, q)
    import bench
    import psgd_torch_amd
    from oracle_engine import OracleEngine
    shapes = bench.gpt2_shapes(n_layer=2, n_embd=64, vocab=4160, block=128)
    world, steps = 8, 3
    g = torch.Generator().manual_seed(7)
    ref = [torch.nn.Parameter(0.5 * torch.randn(s, generator=g)) for s in shapes]
    opt = psgd_torch_amd.KWNS4(ref, preconditioner_dtype=torch.float32, engine_factory=OracleEngine, lr_params=1e-2)
    g = torch.Generator().manual_seed(99)
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)          # (as the workers: the CPU matmul's blocking, hence its rounding, depends on the thread count)
    try:
        for _ in range(steps):
            for p in ref:
                p.grad = 0.3 * torch.randn(p.shape, generator=g)
            opt.step()
    finally:
        torch.set_num_threads(nthreads)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_row_worker, args=(world, _free_port(), d, dict(), shapes, steps), nprocs=world, join=True)
        rs = [torch.load(os.path.join(d, f"r{r}.pt")) for r in range(world)]
    assert rs[0]["split"] == [0], rs[0]["split"]
    assert all(r["load"] >= 1 for r in rs) and sum(r["load"] for r in rs) == len(shapes) - 1 + world
    for k, c in enumerate(ref):
        for r in rs[1:]:
            assert torch.equal(rs[0]["params"][k], r["params"][k]), ("ranks diverged", k)
        if k == 0:
            err = float((rs[0]["params"][k] - c.data).abs().max() / c.data.abs().max())
            assert err <= 2e-6 * steps, ("row-split tensor vs the single-process result", err)
        else:
            # (bitwise in the small-shape tests above; at these sizes the CPU stand-in's fp32 matmuls are not reproducible bit for bit
            #  ACROSS PROCESSES -- the BLAS picks its kernels by operand alignment -- so: one or two ulp)
            err = float((rs[0]["params"][k] - c.data).abs().max() / c.data.abs().max())
            assert err <= 3e-7 * steps, ("unsplit tensor differs from the single-process result", k, err)


def _resync_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        import psgd_torch_amd
        from oracle_engine import OracleEngine
        g = torch.Generator().manual_seed(7)
        params = [torch.nn.Parameter(0.5 * torch.randn(s, generator=g)) for s in ROW_SHAPES]
        opt = psgd_torch_amd.KWNS4(params, preconditioner_dtype=torch.float32, engine_factory=OracleEngine, shard_state=True, lr_params=1e-2,
                                   shard_split_rows=0.0, shard_resync_every=3)
        g = torch.Generator().manual_seed(99)
        seen = []
        for t in range(4):
            for p in params:
                p.grad = 0.3 * torch.randn(p.shape, generator=g)
            opt.step()
            if t == 0 and rank == 1:
                # what a cooperative norm-bound time-out on ONE member leaves behind: that member's replicated dense factor (and its L)
                # differs from its peers'
                for b in opt._buckets.values():
                    for k, i in enumerate(b.owned):
                        if i in b.rows:
                            b.engine.QL(k)[0][1].mul_(1.5)
                            b.engine.QL(k)[1][1].add_(0.25)
                            b.engine.state_changed()
            facs = [(b.engine.QL(k)[0][1].clone(), b.engine.QL(k)[1][1].clone()) for b in opt._buckets.values() for k, i in enumerate(b.owned) if i in b.rows]
            seen.append(facs)
        torch.save(seen, os.path.join(outdir, f"r{rank}.pt"))
    finally:
        torch.distributed.destroy_process_group()


def test_row_split_replicated_factors_are_resynced():
    """The dense factor of a row-split tensor is replicated on every member; a member that falls out of step (a skipped update after a
    norm-bound time-out, drift of unordered atomics) is brought back by the periodic broadcast from member 0 (shard_resync_every)."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_resync_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        rs = [torch.load(os.path.join(d, f"r{r}.pt")) for r in range(world)]
    assert len(rs[0][0]) >= 1
    for t in range(4):
        same = all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(rs[0][t], rs[1][t]))
        # steps 1 and 2 (t = 0, 1 after the perturbation): the members differ; the resync at the end of step 3 (b.step = 3) repairs it
        assert same == (t >= 2), (t, same)
