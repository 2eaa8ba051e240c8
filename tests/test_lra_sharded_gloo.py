"""Row-sharded LRA (last row) on CPU-verifiable terms: psgd_torch_amd.lra_sharded.RowShardedLRA -- the product's phase driver
and exchange -- over a real gloo group (world 2 and 4) with the test-only phase engine (tests/oracle_lra_engine.py), against the
single-process oracle (oracle/psgd_oracle.py: psgd.py:994-1063) on the same inputs.  The ranks must agree bitwise on everything
replicated (the three Lipschitz constants) and, assembled, match the oracle within the arithmetic's rounding; an update issues 4
collectives, an apply 3."""
import os
import socket
import sys
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _inputs(N, r, dt, steps):
    g = torch.Generator().manual_seed(1234)
    U = torch.randn(N, r, generator=g, dtype=torch.float64); V = torch.randn(N, r, generator=g, dtype=torch.float64)
    if r:
        U *= 0.1 ** 0.5 / torch.linalg.vector_norm(U); V *= 0.1 ** 0.5 / torch.linalg.vector_norm(V)
    d = 0.5 + torch.rand(N, 1, generator=g, dtype=torch.float64)
    gs = [torch.randn(N, 1, generator=g, dtype=torch.float64) * (1 + 0.1 * t) for t in range(steps + 1)]
    vs = [torch.randn(N, 1, generator=g, dtype=torch.float64) for _ in range(steps)]
    coins = [0.2, 0.7, 0.4, 0.9, 0.1, 0.6][:steps]
    c = lambda x: x.to(dt).clone()
    return c(U), c(V), c(d), [c(x) for x in gs], [c(x) for x in vs], coins


def _oracle(N, r, dt, steps):
    from oracle import psgd_oracle as O
    U, V, d, gs, vs, coins = _inputs(N, r, dt, steps)
    UVd, Luvd = [U, V, d], [torch.zeros([], dtype=dt) for _ in range(3)]
    for t in range(steps):
        O.update_precond_lra_whiten(UVd, Luvd, gs[t], vs[t], coins[t], lr=0.1, betaL=0.9, damping=1e-9)
    return U, V, d, Luvd, O.precond_grad_lra(UVd, gs[steps])


def _worker(rank, world, port, outdir, N, r, dtname, steps):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle_lra_engine import OracleLraPhaseEngine
        from psgd_torch_amd.lra_sharded import RowShardedLRA, all_gather_rows, shard_rows
        dt = getattr(torch, dtname)
        U, V, d, gs, vs, coins = _inputs(N, r, dt, steps)
        row0, rows = shard_rows(N, world, rank, align=64)
        loc = slice(row0, row0 + rows)
        Luvd = [torch.zeros([], dtype=dt) for _ in range(3)]
        eng = OracleLraPhaseEngine(U[loc].clone(), V[loc].clone(), d[loc].clone(), Luvd)
        drv = RowShardedLRA(eng)
        for t in range(steps):
            drv.update_whiten(gs[t][loc], lr=0.1, betaL=0.9, damping=1e-9, v_noise=vs[t][loc], update_u=coins[t] < 0.5)
        n_update = drv.collectives
        h_loc = drv.precond_grad(gs[steps][loc])
        n_apply = drv.collectives - n_update
        h = all_gather_rows(h_loc, N, world, rank, align=64)
        hsq = float(eng._w("HSQ"))
        torch.save(dict(U=eng.U, V=eng.V, d=eng.d, Luvd=[x.clone() for x in Luvd], h=h, hsq=hsq, row0=row0, rows=rows,
                        n_update=n_update, n_apply=n_apply), os.path.join(outdir, f"r{rank}.pt"))
    finally:
        torch.distributed.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def relerr(a, b):
    return float(torch.linalg.vector_norm(a.double() - b.double()) / (torch.linalg.vector_norm(b.double()) + 1e-300))


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("r,dtname,tol", [(10, "float64", 1e-10), (10, "float32", 2e-4), (3, "float64", 1e-10), (0, "float64", 1e-12)])
def test_row_sharded_lra_matches_the_single_process_oracle(world, r, dtname, tol):
    N, steps = 1000, 4
    dt = getattr(torch, dtname)
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_worker, args=(world, _free_port(), outdir, N, r, dtname, steps), nprocs=world, join=True)
        res = [torch.load(os.path.join(outdir, f"r{k}.pt")) for k in range(world)]
    U0, V0, d0, L0, h0 = _oracle(N, r, dt, steps)
    U = torch.cat([x["U"] for x in res]); V = torch.cat([x["V"] for x in res]); d = torch.cat([x["d"] for x in res])
    assert U.shape == U0.shape and d.shape == d0.shape
    for k in range(1, world):       # replicated scalars and the gathered result: the same bits on every rank
        for a, b in zip(res[0]["Luvd"], res[k]["Luvd"]):
            assert torch.equal(a, b)
        assert torch.equal(res[0]["h"], res[k]["h"])
        assert res[0]["hsq"] == res[k]["hsq"]
    if r:
        assert relerr(U, U0) <= tol and relerr(V, V0) <= tol, (relerr(U, U0), relerr(V, V0))
    assert relerr(d, d0) <= tol, relerr(d, d0)
    for a, b in zip(res[0]["Luvd"], L0):
        assert relerr(a, b) <= tol, (a, b)
    assert relerr(res[0]["h"], h0) <= tol, relerr(res[0]["h"], h0)
    assert abs(res[0]["hsq"] - float((h0.double() ** 2).sum())) <= 10 * tol * float((h0.double() ** 2).sum())
    assert all(x["n_update"] == 4 * steps and x["n_apply"] == 3 for x in res), [(x["n_update"], x["n_apply"]) for x in res]


def test_one_rank_is_the_plain_pipeline():
    """world 1 (no group): the phases back to back are the oracle's update + apply."""
    from oracle_lra_engine import OracleLraPhaseEngine
    from psgd_torch_amd.lra_sharded import RowShardedLRA
    N, r, steps, dt = 300, 7, 3, torch.float64
    U, V, d, gs, vs, coins = _inputs(N, r, dt, steps)
    Luvd = [torch.zeros([], dtype=dt) for _ in range(3)]
    drv = RowShardedLRA(OracleLraPhaseEngine(U.clone(), V.clone(), d.clone(), Luvd))
    for t in range(steps):
        drv.update_whiten(gs[t], lr=0.1, betaL=0.9, damping=1e-9, v_noise=vs[t], update_u=coins[t] < 0.5)
    h = drv.precond_grad(gs[steps])
    U0, V0, d0, L0, h0 = _oracle(N, r, dt, steps)
    assert relerr(drv.engine.U, U0) <= 1e-10 and relerr(drv.engine.V, V0) <= 1e-10 and relerr(drv.engine.d, d0) <= 1e-10
    assert relerr(h, h0) <= 1e-10 and drv.collectives == 0
    for a, b in zip(Luvd, L0):
        assert relerr(a, b) <= 1e-10
