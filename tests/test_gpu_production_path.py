"""Oracle parity of the PRODUCTION noise path on the PRODUCTION plans (-m gpu).

Every other parity test feeds explicit noise buffers (`opt._replay` / `noise=`).  What `bench.py` times -- and what a user runs --
is the other branch: `_replay is None`, the damped input written by psgdk_accumulate's fused `damp` branch with in-kernel Philox
draws (kernels_ew.hiph: tile_noise), the 32 x d start blocks of both norm bounds drawn inside nlb_coop_kernel / nlb_init_kernel
(kernels_dense.hiph: nlb_philox4), host gates from KWNS4's own generator.  Here that branch runs UNTOUCHED (no hook installed on
the optimizer); afterwards `psgdk_test_dump_noise` (include/psgdk_test.h) writes out what those kernels drew for the same
(seed, offset) -- it calls the same device functions in the same tile mapping -- and the draws are replayed into
`orc.KWNS4Oracle`, whose host gate stream is re-created from the optimizer's seed.  Compared per tensor: the parameter update, P =
Q^T Q, L and the EMA, with the criteria of the replay tests (fp32: 3e-5 per step; bf16: error vs the fp64 oracle trajectory <= 1.5 x
the oracle-bf16's own error + a floor of one or two bf16 ulp).

Plans: the full 148-tensor GPT-2-small parameter list (misc/gpt2.py:116-118,187-189,215-227,238-254) -- exactly `bench.py`'s
configuration: XCD queue cutting over mixed problems, the split-K `wte` Gram inside the batch, the 256 x 256 tiling picked by tile
count, the cooperative norm bound at 186 workgroups -- on both norm-bound routes, with the balancing step (psgd.py:418-419) firing on
at least one matrix; the 292-tensor GPT-2-medium list (1024-wide factors: multi-launch route); one block in fp32.
Match: psgd.py:402-403,62,87; wrapped_as_torch_optimizer_for_ddp.py:98-176."""
import os

import pytest
import torch

from helpers import P_of, relerr
from oracle import psgd_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ULP = 7.8125e-3


def gpt2_shapes(d=768, layers=12, vocab=50304, ctx=1024):
    """The parameter list of misc/gpt2.py's GPT (wte tied to the head): wte, wpe, per block ln_1 (w, b), c_attn (w, b), attn c_proj
    (w, b), ln_2 (w, b), c_fc (w, b), mlp c_proj (w, b), then ln_f (w, b).  The same list bench.py builds."""
    s = [(vocab, d), (ctx, d)]
    for _ in range(layers):
        s += [(d,), (d,), (3 * d, d), (3 * d,), (d, d), (d,), (d,), (d,), (4 * d, d), (4 * d,), (d, 4 * d), (d,)]
    return s + [(d,), (d,)]


def _structured(shape, seed, scale=0.3):
    """g = H1 V H2 with SPD H of condition ~1e3, so that the preconditioner has something to fit."""
    g = torch.Generator().manual_seed(seed)
    m, n = shape
    V = torch.randn(m, n, generator=g)
    sm = torch.logspace(0, -1.5, m).sqrt()[torch.randperm(m, generator=g)]
    sn = torch.logspace(0, -1.5, n).sqrt()[torch.randperm(n, generator=g)]
    return scale * sm[:, None] * V * sn[None, :]


def _seed_with_a_balancing_matrix(shapes, steps, start=0):
    """The first optimizer seed >= start whose host gate stream (one group gate, then one balancing gate per tensor, per step) fires
    the 1 % balancing on at least one tensor with two factors within `steps` steps."""
    two = [len([x for x in s if x != 1]) == 2 for s in shapes]
    for seed in range(start, start + 2000):
        g = torch.Generator().manual_seed(seed)
        for _ in range(steps):
            torch.rand([], generator=g)
            u = [float(torch.rand([], generator=g)) for _ in shapes]
            if any(t and x < 0.01 for t, x in zip(two, u)):
                return seed
    raise AssertionError("no seed found")


def _production_vs_oracle(shapes, kw, steps, seed, grad_seed=0):
    import psgd_torch_amd as amd
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 32)))
    gen = torch.Generator().manual_seed(1000 + grad_seed)
    p_cpu = [0.02 * torch.randn(s, generator=gen) for s in shapes]
    params = [torch.nn.Parameter(p.clone().to(DEV)) for p in p_cpu]
    opt = amd.KWNS4(params, seed=seed, **kw)
    assert opt._replay is None
    gate = torch.Generator().manual_seed(seed)                 # replica of KWNS4's host gate generator (KWNS4._gate_gen)
    queue, queue64 = [], []
    first = kw.get("update_preconditioner_first", True)

    ora = orc.KWNS4Oracle([p.clone() for p in p_cpu], uniform=lambda: queue.pop(0), noise_for=lambda G, kinds: queue.pop(0), **kw)
    kw64 = dict(kw)
    ora64 = orc.KWNS4Oracle([p.double() for p in p_cpu], uniform=lambda: queue64.pop(0), noise_for=lambda G, kinds: queue64.pop(0), **kw64)
    ora64.g["preconditioner_dtype"] = torch.float64
    balanced = 0
    for t in range(steps):
        grads = [_structured(s, 7919 * t + 31 * i + grad_seed) if len(s) == 2 else 0.3 * torch.randn(s, generator=gen)
                 for i, s in enumerate(shapes)]
        for p, g in zip(params, grads):
            p.grad = g.to(DEV)
        opt.step()                                              # the production path: Philox in the kernels, own gate generator
        (bucket,) = opt._buckets.values()
        eng = bucket.engine
        # what it drew: host gates from the replica generator, device noise from the dump hook
        u_group = float(torch.rand([], generator=gate))
        assert u_group < kw.get("preconditioner_update_probability", 1.0)
        u_bal = [float(torch.rand([], generator=gate)) for _ in shapes]
        g_nz, spd, skh = eng.dump_noise(seed, 2 * t + (0 if first else 1))
        per = []
        for i in range(len(shapes)):
            nf = len(eng.kinds[i])
            per.append(orc.KronNoise(g_nz[i].cpu().reshape(tuple(x for x in shapes[i] if x != 1)),
                                     [spd[(i, j)].cpu() if (i, j) in spd else None for j in range(nf)],
                                     [skh[(i, j)].cpu() if (i, j) in skh else None for j in range(nf)], u_bal[i]))
            balanced += int(u_bal[i] < 0.01 and nf == 2)
        pdt = kw.get("preconditioner_dtype", torch.bfloat16)
        queue[:] = [u_group] + [orc.KronNoise(n.g_noise.to(pdt), n.spd, n.skh, n.balance_u) for n in per]
        queue64[:] = [u_group] + [orc.KronNoise(n.g_noise.double(), [x.double() if x is not None else None for x in n.spd],
                                                [x.double() if x is not None else None for x in n.skh], n.balance_u) for n in per]
        ora.step([g.clone() for g in grads])
        ora64.step([g.double() for g in grads])
        assert not queue and not queue64
    torch.cuda.synchronize()
    rows = []
    for i, (p, q, q64, p0) in enumerate(zip(params, ora.params, ora64.params, p_cpu)):
        assert bool(torch.isfinite(p).all())
        st = opt.state[p]
        d64 = q64 - p0.double()
        row = {"i": i, "shape": shapes[i],
               "dp": (relerr(p.detach().cpu().double() - p0.double(), d64), relerr(q.double() - p0.double(), d64)), "P": [], "L": []}
        for j in range(len(st["QL"][0])):
            truthP = P_of([ora64.state[i]["QL"][0][j]])[0]
            row["P"].append((relerr(P_of([st["QL"][0][j]])[0], truthP), relerr(P_of([ora.state[i]["QL"][0][j]])[0], truthP)))
            truthL = ora64.state[i]["QL"][1][j]
            row["L"].append((relerr(st["QL"][1][j], truthL), relerr(ora.state[i]["QL"][1][j], truthL)))
        if ora.state[i]["ema"] is not None:
            t64 = ora64.state[i]["ema"].reshape(-1)
            row["ema"] = (relerr(st["ema"].reshape(-1), t64), relerr(ora.state[i]["ema"].reshape(-1), t64))
        rows.append(row)
    return opt, rows, balanced


def _check_bf16(rows):
    for r in rows:
        e_hip, e_ref = r["dp"]
        assert e_hip <= 1.5 * e_ref + ULP, ("dp", r["i"], r["shape"], e_hip, e_ref)
        for j, (e_hip, e_ref) in enumerate(r["P"]):
            assert e_hip <= 1.5 * e_ref + 1e-2, ("P", r["i"], r["shape"], j, e_hip, e_ref)
        for j, (e_hip, e_ref) in enumerate(r["L"]):
            assert e_hip <= 1.5 * e_ref + 2 * ULP, ("L", r["i"], r["shape"], j, e_hip, e_ref)
        if "ema" in r:
            assert r["ema"][0] <= 1.5 * r["ema"][1] + ULP, ("ema", r["i"], r["shape"], r["ema"])


@pytest.mark.parametrize("fused", [True, False])
def test_gpt2_small_full_plan_production_noise_bf16(fused, monkeypatch):
    """bench.py's configuration, literally: 148 tensors, bf16 preconditioner, fp32 parameters, KWNS4 defaults, 2 steps."""
    if not fused:
        monkeypatch.setenv("PSGDK_NLB_FUSED", "0")
    shapes = gpt2_shapes()
    assert len(shapes) == 148 and sum(int(torch.Size(s).numel()) for s in shapes) == 124_475_904
    seed = _seed_with_a_balancing_matrix(shapes, 2)
    opt, rows, balanced = _production_vs_oracle(shapes, dict(lr_params=1e-3), steps=2, seed=seed)
    assert balanced >= 1
    _check_bf16(rows)
    info = next(iter(opt._buckets.values())).engine.info()
    assert info["nlb_coop"] == (1 if fused else 0) and info["nlb_fallbacks"] == 0 and info["dense_factors"] == 62, info


def test_gpt2_medium_full_plan_production_noise_bf16():
    """The 292-tensor GPT-2-medium list (354,871,296 parameters; 123 dense factors of 1024: multi-launch norm-bound route, the
    256 x 256 tiling on the full-size products), 1 step."""
    shapes = gpt2_shapes(d=1024, layers=24)
    assert len(shapes) == 292 and sum(int(torch.Size(s).numel()) for s in shapes) == 354_871_296
    seed = _seed_with_a_balancing_matrix(shapes, 1)
    opt, rows, balanced = _production_vs_oracle(shapes, dict(lr_params=1e-3), steps=1, seed=seed)
    assert balanced >= 1
    _check_bf16(rows)
    info = next(iter(opt._buckets.values())).engine.info()
    assert info["nlb_coop"] == 0 and info["dense_factors"] == 123 and info["max_dense_dim"] == 1024, info


@pytest.mark.parametrize("kw", [dict(), dict(whiten_grad=True, update_preconditioner_first=False, weight_decay=0.01,
                                             decoupled_weight_decay=False)])
def test_gpt2_block_production_noise_fp32(kw):
    """wpe + one block with an fp32 preconditioner (768-wide fp32 factors: multi-launch route), production noise path, 3 steps:
    3e-5 per step against the fp64 oracle (the fp32 oracle's own error is reported alongside).  Second case: the update AFTER the
    apply (offset 2t + 1), whitening the gradient, coupled weight decay."""
    shapes = [(1024, 768), (768,), (768,), (2304, 768), (2304,), (768, 768), (768,), (768,), (768,), (3072, 768), (3072,),
              (768, 3072), (768,)]
    steps = 3
    seed = _seed_with_a_balancing_matrix(shapes, steps)
    opt, rows, balanced = _production_vs_oracle(shapes, dict(preconditioner_dtype=torch.float32, lr_params=1e-3, **kw), steps=steps, seed=seed)
    assert balanced >= 1
    tol = 3e-5 * steps
    for r in rows:
        assert r["dp"][0] <= tol, ("dp", r["i"], r["shape"], r["dp"])
        for j, (e_hip, e_ref) in enumerate(r["P"]):
            assert e_hip <= tol, ("P", r["i"], r["shape"], j, e_hip, e_ref)
        for j, (e_hip, e_ref) in enumerate(r["L"]):
            assert e_hip <= tol, ("L", r["i"], r["shape"], j, e_hip, e_ref)
        if "ema" in r:
            assert r["ema"][0] <= 1e-6, ("ema", r["i"], r["ema"])


def test_dumped_noise_drives_the_explicit_route_to_the_same_state():
    """The dump hook against the production kernels themselves, without any oracle: engine A runs the production branch (damped
    input fused into psgdk_accumulate, Philox inside the kernels), engine B is fed the DUMPED draws through the explicit-noise
    route (make_x_kernel + caller start blocks).  Both must reach the same Q and L -- up to the order of the fp32 row-sum atomics
    (two runs of the same route differ in the last bits, profiles/r02_experiments/nlb_route_vs_route.txt); draws that were not the
    ones the kernels used would show at the 1e-2 (bf16: eps |S| damping) / 1e-3 (damping) level, and in L directly."""
    import psgd_torch_amd as amd
    from psgd_torch_amd import _lib as L
    shapes = [(192, 64), (64, 64), (40, 130), (33,), (6, 5, 3, 3), (257, 96), ()]
    for dt, tol in ((torch.bfloat16, 2e-3), (torch.float32, 1e-5)):
        engs = [amd.engine.KronEngine(shapes, DEV, precond_dtype=dt, tensor_ids=[3, 1, 4, 5, 9, 2, 6]) for _ in range(2)]
        gen = torch.Generator().manual_seed(5)
        for step in range(3):
            grads = [torch.randn(s, generator=gen).to(DEV) for s in shapes]
            seed, off = 1234, 2 * step
            for e, philox in zip(engs, (True, False)):
                e.accumulate(grads, beta=0.5, damp=dict(source=L.SRC_EMA, damping=1e-3, seed=seed, offset=off) if philox else None)
                noise = None if philox else e.dump_noise(seed, off)
                e.update_precond(L.SRC_EMA, 0.5, 0.9, 1e-3, seed=seed, offset=off, noise=noise, balance_mask=[step == 1] * len(shapes))
            torch.cuda.synchronize()
            for t in range(len(shapes)):
                for qa, qb in zip(engs[0].Q[t], engs[1].Q[t]):
                    assert relerr(qa.float(), qb.float()) <= tol, (dt, step, t, relerr(qa.float(), qb.float()))
                for la, lb in zip(engs[0].Lip[t], engs[1].Lip[t]):
                    assert relerr(la, lb) <= tol, (dt, step, t)
        # and a different offset does give different draws (the comparison above is not vacuous)
        a, b = engs[0].dump_noise(1234, 0), engs[0].dump_noise(1234, 2)
        assert not torch.equal(a[0][0], b[0][0]) and not torch.equal(a[1][(0, 1)], b[1][(0, 1)])
        assert not torch.equal(a[1][(0, 1)], a[2][(0, 1)])          # spd and skh start blocks are different streams
        assert abs(float(a[0][0].float().std()) - 1.0) < 0.05 and abs(float(a[0][0].float().mean())) < 0.05
