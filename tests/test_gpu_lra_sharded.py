"""Row-sharded LRA with the REAL HIP engine (include/psgdk.h "row shards of ONE LRA preconditioner"):
 * the phases back to back on one rank are the one-call update / apply (same launches, same order), every rank class;
 * two ranks sharing cuda:0 over gloo (RCCL refuses two ranks on one device; the transport is not what is under test): the functional
   seam with the engine's own Philox noise (the shards must draw what one GPU draws: counters run over the whole vector) and
   LRAWhiten(shard_rows=True) against the unsharded optimizer -- ranks bitwise equal, results within fp32 rounding of one GPU's."""
import os
import socket
import sys
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def relerr(a, b):
    return float(torch.linalg.vector_norm(a.double().cpu() - b.double().cpu()) / (torch.linalg.vector_norm(b.double().cpu()) + 1e-300))


def _state(N, r, dt, seed=5):
    g = torch.Generator().manual_seed(seed)
    U = torch.randn(N, r, generator=g); V = torch.randn(N, r, generator=g)
    if r:
        U *= 0.1 ** 0.5 / torch.linalg.vector_norm(U); V *= 0.1 ** 0.5 / torch.linalg.vector_norm(V)
    d = 0.5 + torch.rand(N, 1, generator=g)
    gs = [torch.randn(N, 1, generator=g) for _ in range(4)]
    vs = [torch.randn(N, 1, generator=g) for _ in range(3)]
    return U.to(dt), V.to(dt), d.to(dt), [x.to(dt) for x in gs], [x.to(dt) for x in vs]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("r", [0, 10, 24, 48, 80, 130])      # (80, 130: the general path above rank 64, cut into the same phases in round 6)
@pytest.mark.parametrize("philox", [False, True])
def test_phases_back_to_back_are_the_one_call_forms(dt, r, philox):
    from psgd_torch_amd import lra
    from psgd_torch_amd.lra_sharded import RowShardedLRA
    N = 4000 + 37
    U, V, d, gs, vs = _state(N, r, dt)

    def fresh():
        UVd = [U.clone().to(DEV), V.clone().to(DEV), d.clone().to(DEV)]
        L3 = torch.zeros(3, dtype=torch.float32, device=DEV)
        return UVd, L3, lra._LraEngine(UVd, L3)
    UVd_a, L_a, eng_a = fresh()
    UVd_b, L_b, eng_b = fresh()
    eng_b.set_row_shard(0)
    drv = RowShardedLRA(eng_b)
    for t in range(3):
        kw = dict(seed=11 + t, offset=t) if philox else dict(v_noise=vs[t].to(DEV))
        eng_a.update_whiten(gs[t].to(DEV), 0.1, 0.9, 1e-9, update_u=(t % 2 == 0), **kw)
        drv.update_whiten(gs[t].to(DEV), lr=0.1, betaL=0.9, damping=1e-9, update_u=(t % 2 == 0), **kw)
    ha = eng_a.precond_grad(gs[3].to(DEV))
    hb = drv.precond_grad(gs[3].to(DEV))
    torch.cuda.synchronize()
    # same launches in the same order; the row passes add their partial sums with fp32 atomics whose order is not fixed, so two runs of
    # EITHER form differ in the last bits: rounding-level agreement, not bitwise
    tol = 1e-5 if dt == torch.float32 else 2e-2
    for a, b in zip(UVd_a, UVd_b):
        assert a.numel() == 0 or relerr(a, b) <= tol, relerr(a, b)
    assert relerr(L_a, L_b) <= tol and relerr(ha, hb) <= tol
    assert torch.isfinite(ha.float()).all()
    # a declared shard refuses the one-call forms (its reductions would be incomplete)
    with pytest.raises(Exception):
        eng_b.update_whiten(gs[0].to(DEV), 0.1, 0.9, 1e-9, v_noise=vs[0].to(DEV))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _functional_worker(rank, world, port, outdir, N, r, dt=torch.float32):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from psgd_torch_amd import lra
        from psgd_torch_amd.lra_sharded import RowShardedLRA, all_gather_rows, shard_rows
        U, V, d, gs, _ = _state(N, r, dt)
        row0, rows = shard_rows(N, world, rank)
        loc = slice(row0, row0 + rows)
        UVd = [U[loc].clone().to(DEV), V[loc].clone().to(DEV), d[loc].clone().to(DEV)]
        L3 = torch.zeros(3, dtype=torch.float32, device=DEV)
        eng = lra._LraEngine(UVd, L3)
        eng.set_row_shard(row0)
        drv = RowShardedLRA(eng)
        for t in range(3):
            drv.update_whiten(gs[t][loc].to(DEV), lr=0.1, betaL=0.9, damping=1e-9, seed=11 + t, offset=t, update_u=(t % 2 == 0))
        h = all_gather_rows(drv.precond_grad(gs[3][loc].to(DEV)), N, world, rank)
        torch.cuda.synchronize()
        torch.save(dict(U=UVd[0].cpu(), V=UVd[1].cpu(), d=UVd[2].cpu(), L=L3.cpu(), h=h.cpu(), n=drv.collectives, packed=eng.info()["packed_rows"], rows=rows),
                   os.path.join(outdir, f"r{rank}.pt"))
    finally:
        torch.distributed.destroy_process_group()


def test_two_ranks_one_gpu_bf16_packed_row_kernels():
    """Round 6: the shards of a bf16 preconditioner of even rank <= 16 run the packed two-rows-per-thread kernels (kernels_lra_pk.hiph) over
    their whole blocks, each with its own row0: the Philox counters of the damping noise run over the WHOLE vector, so a shard starting at an
    odd multiple of anything must draw what one GPU draws for those rows (a wrong counter is an O(1) error in U, V, d).  Two ranks on one
    GPU against the unsharded engine, bf16: agreement to the rounding of bf16 stores."""
    from psgd_torch_amd import lra
    N, world, r = 9000 + 91, 2, 10
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_functional_worker, args=(world, _free_port(), outdir, N, r, torch.bfloat16), nprocs=world, join=True)
        res = [torch.load(os.path.join(outdir, f"r{k}.pt")) for k in range(world)]
    for x in res:
        assert x["packed"] == x["rows"] // 512 * 512 > 0, (x["packed"], x["rows"])
    U, V, d, gs, _ = _state(N, r, torch.bfloat16)
    UVd = [U.to(DEV), V.to(DEV), d.to(DEV)]
    L3 = torch.zeros(3, dtype=torch.float32, device=DEV)
    eng = lra._LraEngine(UVd, L3)
    for t in range(3):
        eng.update_whiten(gs[t].to(DEV), 0.1, 0.9, 1e-9, seed=11 + t, offset=t, update_u=(t % 2 == 0))
    h = eng.precond_grad(gs[3].to(DEV))
    torch.cuda.synchronize()
    assert eng.info()["packed_rows"] == N // 512 * 512
    assert torch.equal(res[0]["L"], res[1]["L"]) and torch.equal(res[0]["h"], res[1]["h"])
    for k, nm in enumerate(("U", "V", "d")):
        got = torch.cat([x[nm] for x in res])
        assert relerr(got, UVd[k]) <= 1e-2, (nm, relerr(got, UVd[k]))
    assert relerr(res[0]["L"], L3) <= 1e-2 and relerr(res[0]["h"], h) <= 1e-2


@pytest.mark.parametrize("r", [10, 24, 80])
def test_two_ranks_one_gpu_match_one_gpu_with_the_engines_own_noise(r):
    from psgd_torch_amd import lra
    N, world = 6000 + 91, 2
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_functional_worker, args=(world, _free_port(), outdir, N, r), nprocs=world, join=True)
        res = [torch.load(os.path.join(outdir, f"r{k}.pt")) for k in range(world)]
    U, V, d, gs, _ = _state(N, r, torch.float32)
    UVd = [U.to(DEV), V.to(DEV), d.to(DEV)]
    L3 = torch.zeros(3, dtype=torch.float32, device=DEV)
    eng = lra._LraEngine(UVd, L3)
    for t in range(3):
        eng.update_whiten(gs[t].to(DEV), 0.1, 0.9, 1e-9, seed=11 + t, offset=t, update_u=(t % 2 == 0))
    h = eng.precond_grad(gs[3].to(DEV))
    torch.cuda.synchronize()
    assert torch.equal(res[0]["L"], res[1]["L"]) and torch.equal(res[0]["h"], res[1]["h"])
    assert res[0]["n"] == 3 * 4 + 3
    for k, nm in enumerate(("U", "V", "d")):
        got = torch.cat([x[nm] for x in res])
        assert relerr(got, UVd[k]) <= 2e-5, (nm, relerr(got, UVd[k]))       # (a wrong Philox counter would be an O(1) error)
    assert relerr(res[0]["L"], L3) <= 2e-5 and relerr(res[0]["h"], h) <= 2e-5


SHAPES = [(40, 30), (300,), (25, 64)]


def _whiten_run(shard, dev, steps=4):
    from psgd_torch_amd import lra
    g = torch.Generator().manual_seed(3)
    params = [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in SHAPES]
    N = sum(p.numel() for p in params)
    r = 4
    opt = lra.LRAWhiten(params, rank_of_approximation=r, preconditioner_init_scale=1.0, lr_params=1e-2, lr_preconditioner=0.1, momentum=0.9,
                        shard_rows=shard, seed=17)
    U0 = torch.randn(N, r, generator=g); U0 *= 0.1 ** 0.5 / torch.linalg.vector_norm(U0)
    V0 = torch.randn(N, r, generator=g); V0 *= 0.1 ** 0.5 / torch.linalg.vector_norm(V0)
    sh = opt._shard
    loc = slice(sh["row0"], sh["row0"] + sh["rows"]) if sh else slice(0, N)
    opt._UVd[0].copy_(U0[loc]); opt._UVd[1].copy_(V0[loc])
    draws = torch.rand(3 * steps, generator=g).tolist()
    noise = [torch.randn(N, 1, generator=g) for _ in range(steps)]
    cs = [[torch.randn(s, generator=g).to(dev) for s in SHAPES] for _ in range(steps)]
    for t in range(steps):
        u = iter(draws[3 * t:3 * t + 3])
        opt._uniform = lambda: next(u)
        opt._v_noise = lambda t=t: noise[t].to(dev)
        c = cs[t]
        opt.step(lambda: sum((p * p * x).sum() for p, x in zip(params, c)))
    torch.cuda.synchronize()
    return [p.data.cpu() for p in params]


def _whiten_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.save(_whiten_run(True, DEV), os.path.join(outdir, f"r{rank}.pt"))
    finally:
        torch.distributed.destroy_process_group()


def test_lrawhiten_shard_rows_two_ranks_one_gpu():
    world = 2
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_whiten_worker, args=(world, _free_port(), outdir), nprocs=world, join=True)
        res = [torch.load(os.path.join(outdir, f"r{k}.pt")) for k in range(world)]
    ref = _whiten_run(False, DEV)
    for a, b, c in zip(res[0], res[1], ref):
        assert torch.equal(a, b)                      # every rank applies the identical update
        assert relerr(a, c) <= 1e-5, relerr(a, c)
