"""
The cooperative spectral-norm bound (nlb_coop_kernel: norm_lower_bound_spd/_skh, psgd.py:46-93, in ONE launch per bound with the
S workgroups of a factor exchanging their slabs inside the launch) under conditions that break a placement- or timing-dependent
exchange: thousands of launches while a second stream keeps every CU busy with unrelated work (siblings start at different
times, on whatever CUs free up), L1-warm consumers (the same buffers are re-read launch after launch), every word checked
against the multi-launch route of the same arithmetic.  Plus the failure path: a sibling that never arrives must surface as
PSGDK_ERR_NLB_TIMEOUT / a warning, leave the state valid and move the plan to the multi-launch route.
"""
import ctypes as C
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _engine(n_factors, width, dt):
    import psgd_torch_amd as amd
    from psgd_torch_amd import _lib as L
    # (4*width, width): the long dim is diagonal (max_skew 1), the short one a dense factor -- GPT-2-small's c_fc shape at 768
    eng = amd.KronEngine([(4 * width, width)] * n_factors, DEV, precond_dtype=dt, use_momentum=False, init_scale=1.0)
    g = torch.Generator().manual_seed(1)
    grads = [(0.3 * torch.randn(4 * width, width, generator=g)).to(dt).to(DEV) for _ in range(n_factors)]
    # Q = I gives an exactly symmetric Q' and hence R = Q'^T - Q' = 0: start from a visibly non-symmetric Q so that the skh chain
    # (A = R) iterates on real data as well
    for t in range(n_factors):
        q = eng.Q[t][1]
        assert q.dim() == 2
        q.add_((0.05 * torch.randn(width, width, generator=g)).to(dt).to(DEV))
    eng.state_changed()
    eng.accumulate(grads, keep_grad=True)
    eng.update_precond(L.SRC_GRAD, 0.5, 0.9, 1e-9, seed=3, offset=0, balance_mask=[False] * n_factors)   # fills term1, R, row stats
    torch.cuda.synchronize()
    return eng, grads


def _run_bound(eng, chain, route, seed, vsq, v, fault=0):
    from psgd_torch_amd import _lib as L
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(eng.lib.psgdk_test_nlb(eng._plan, chain, route, seed, 0, vsq.data_ptr(), v.data_ptr(), fault, st), "test_nlb")


@pytest.mark.parametrize("dt,width,n_factors,iters", [(torch.bfloat16, 768, 62, 1000), (torch.float32, 384, 40, 300),
                                                       (torch.bfloat16, 320, 24, 300), (torch.bfloat16, 768, 16, 300),
                                                       (torch.bfloat16, 1024, 16, 200), (torch.float32, 512, 12, 100)])
def test_cooperative_bound_soak_under_load(dt, width, n_factors, iters):
    """GPT-2-small's plan (62 factors x 768, S = 3) and two other widths: 2 chains x iters bounds with fresh noise each, a
    saturating side stream (siblings start at different times), consumers L1-warm by construction (the two exchange buffers
    are re-read every second product of a launch), each launch compared ELEMENT BY ELEMENT with the multi-launch route on the
    same inputs -- the last two blocks of the iteration, all four products' row sums.
    What a correct exchange may differ by: the routes share MFMA, K order and rounding points, but accumulate the fp32 row sums
    with atomics, whose order is free -- in BOTH routes: two runs of the SAME route differ as much as the two routes do
    (tools/nlb_diag.py: row sums 1e-4 in bf16 / 5e-7 in fp32, a few hundred elements of V by one ulp).  A 1-ulp-of-fp32 change of
    a row scale flips the rounding of some elements of the next block by one ulp of the element type.
    What a stale word would look like: two bf16 elements of a block replaced by the previous contents of that exchange buffer
    (the block of two products earlier): an O(1) error on them, and ~ 1/sqrt(d) = 4 % of a typical element on 256 elements of
    the next block.  Bound: no element off by more than 2 ulp of its row's largest element."""
    eng, _ = _engine(n_factors, width, dt)
    info = eng.info()
    assert info["nlb_coop"] == 1, info
    # (round 6) plans whose members of 128 columns fit the CUs take them: 62 x 768 needs 372 workgroups and keeps members of 256 (S = 3),
    # 16 x 768 runs six members per factor, 40 x 384 fp32 and 24 x 320 three; 16 x 1024 bf16 (GPT-2-medium's width: a rank's share of an 8-way
    # sharded job) and 12 x 512 fp32 run eight / four members with 32 K steps of registers -- widths the cooperative launch did not cover before
    # (second step of round 6: 62 x 768 takes members of 192 columns -- twelve waves --, four per factor, 248 workgroups)
    assert info["nlb_member_cols"] == (192 if n_factors == 62 else 128), info
    F, dp = info["dense_factors"], info["max_dense_dim"]
    vsq_ref = torch.zeros(F, 4, 32, device=DEV)
    v_ref = torch.zeros(F, 2, 32, dp, device=DEV, dtype=dt)
    vsq = torch.zeros_like(vsq_ref)
    v = torch.zeros_like(v_ref)
    # unrelated work on a second stream: large matmuls that keep all CUs occupied and make the siblings' start times uneven
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=DEV, dtype=torch.bfloat16)
    worst_v = torch.zeros((), device=DEV)
    worst_s = torch.zeros((), device=DEV)
    nonzero = torch.zeros((), device=DEV)
    ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -22          # of the row's largest element (its binade's upper end)
    for chain in (0, 1):
        for it in range(iters):
            if it % 4 == 0:
                with torch.cuda.stream(side):
                    for _ in range(3):
                        a = (a @ a).clamp_(-1, 1)
            seed = 100000 * chain + it
            _run_bound(eng, chain, 0, seed, vsq_ref, v_ref)
            _run_bound(eng, chain, 1, seed, vsq, v)
            rowmax = v_ref.float().abs().amax(dim=-1, keepdim=True)
            dv = ((v.float() - v_ref.float()).abs() / rowmax.clamp_min(1e-30)).amax()
            ds = ((vsq - vsq_ref).abs() / vsq_ref.abs().clamp_min(1e-30)).amax()
            worst_v = torch.maximum(worst_v, dv)
            worst_s = torch.maximum(worst_s, ds)
            nonzero = torch.maximum(nonzero, rowmax.amin())
    torch.cuda.synchronize()
    assert bool(torch.isfinite(v_ref.float()).all()) and float(nonzero) > 0      # (every row of every block carried data)
    # (fp32: a 1-ulp change of a row scale moves EVERY element of the next block by up to an ulp, and two products follow it:
    #  a few ulp in total; still four orders of magnitude below what a stale word would do)
    assert float(worst_v) <= (2 * ulp if dt == torch.bfloat16 else 8 * ulp), (float(worst_v), ulp)
    assert float(worst_s) <= (1e-2 if dt == torch.bfloat16 else 2e-5), float(worst_s)
    assert eng.info()["nlb_fallbacks"] == 0 and eng.info()["nlb_coop"] == 1


def test_cooperative_bound_timeout_is_reported_and_falls_back():
    """A sibling that never announces its slab: the waiting workgroups give up after the spin limit, the factor's step is
    skipped (mu = 0 / s = 0: state stays finite), the host-mapped error word is set, the next update call returns
    PSGDK_ERR_NLB_TIMEOUT exactly once -- the engine warns and repeats the call on the multi-launch route."""
    from psgd_torch_amd import _lib as L
    eng, grads = _engine(8, 768, torch.bfloat16)
    assert eng.info()["nlb_coop"] == 1
    F, dp = eng.info()["dense_factors"], eng.info()["max_dense_dim"]
    vsq = torch.zeros(F, 4, 32, device=DEV)
    v = torch.zeros(F, 2, 32, dp, device=DEV, dtype=torch.bfloat16)
    _run_bound(eng, 0, 1, 5, vsq, v, fault=1)
    torch.cuda.synchronize()          # (the word is written by the kernel; the library itself never synchronises for it)
    q_before = [q.clone() for q in eng.Q[0]]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        eng.accumulate(grads, keep_grad=True)
        eng.update_precond(L.SRC_GRAD, 0.5, 0.9, 1e-9, seed=4, offset=0, balance_mask=[False] * 8)
    assert any("timed out" in str(x.message) for x in w), [str(x.message) for x in w]
    info = eng.info()
    assert info["nlb_coop"] == 0 and info["nlb_fallbacks"] == 1, info
    torch.cuda.synchronize()
    for t in range(8):
        for q in eng.Q[t]:
            assert bool(torch.isfinite(q.float()).all())
    assert any(not torch.equal(a, b) for a, b in zip(q_before, eng.Q[0])), "the repeated call must have updated Q"
    # and the next call is silent
    with warnings.catch_warnings(record=True) as w2:
        warnings.simplefilter("always")
        eng.accumulate(grads, keep_grad=True)
        eng.update_precond(L.SRC_GRAD, 0.5, 0.9, 1e-9, seed=5, offset=0, balance_mask=[False] * 8)
    assert not w2


@pytest.mark.parametrize("pd", [torch.float32, torch.bfloat16])
def test_small_plans_are_bitwise_reproducible_and_plan_independent(pd):
    """Replicas of a parameter (DTensor Replicate placements, DDP replicas) sit in engines of DIFFERENT composition on different ranks
    and are resynced only now and then (..._dtensor.py:168-179): for factors that ONE 64 x 64 block covers every launch must be a pure
    function of that tensor's own inputs -- no sum whose order depends on wave arrival, nothing that depends on what else is in the plan.
    (Wider factors add three or more blocks' shares of a trace or a row sum with device atomics, like the reference's own kernels -- the reason
    it resyncs at all, ..._ddp.py:163.)
    (Round 4 found the narrow-factor instantiation of the cooperative bound adding four waves' row sums with LDS atomics: 1-ulp differences
    of mu from run to run, visible in fp32 only.)  Three optimizers over clones of the same tensors: A, A again, and B whose second tensor
    never has a gradient (one tensor fewer in its engine; same positions, hence the same Philox streams); 12 trials x 6 steps, all
    states compared bit for bit.  First GPU run: A against A identical in all 3 x 12 trials after the fix; A against B differed where the
    balancing gate had picked different tensors (the draws go one per tensor with a gradient) -- the gate is pinned here."""
    import psgd_torch_amd
    shapes = [(96, 64), (1, 80), (96,), (64, 64), (65, 40)]
    bad = []
    for trial in range(12):
        g = torch.Generator().manual_seed(100 + trial)
        init = [0.5 * torch.randn(s, generator=g) for s in shapes]
        grads = [[0.3 * torch.randn(s, generator=g).to(DEV) for s in shapes] for _ in range(6)]
        runs = []
        for variant in ("A", "A", "B"):
            ps = [torch.nn.Parameter(x.clone().to(DEV)) for x in init]
            opt = psgd_torch_amd.KWNS4(ps, preconditioner_dtype=pd, lr_params=1e-2, seed=trial)
            # the host's gate draws (update: always at probability 1; balancing: 1 % per tensor) come one per tensor WITH a gradient, so the
            # two compositions would balance different tensors on different steps -- host logic, not what is under test
            opt._uniform = lambda: 0.5
            for gs in grads:
                for i, (p, gr) in enumerate(zip(ps, gs)):
                    p.grad = None if (variant == "B" and i == 1) else gr
                opt.step()
            torch.cuda.synchronize()
            st = {}
            for i, p in enumerate(ps):
                if variant == "B" and i == 1:
                    continue
                Q, Ls = opt.state[p]["QL"]
                st[i] = [p.detach().clone()] + [q.clone() for q in Q] + [ell.clone() for ell in Ls] + [opt.state[p]["ema"].clone()]
            runs.append(st)
        for other, what in ((runs[1], "the same plan run twice"), (runs[2], "a plan with one tensor fewer")):
            for i, ts in other.items():
                for k, (a, b) in enumerate(zip(runs[0][i], ts)):
                    if not torch.equal(a, b):
                        bad.append(f"trial {trial}: tensor {i} {shapes[i]}, state item {k}: {what} differs by "
                                   f"{float((a.float() - b.float()).abs().max()):.3e}")
    assert not bad, "\n".join(bad[:40])


def test_narrow_and_wide_members_agree_with_the_multi_launch_route(monkeypatch):
    """Round 6: a plan with few wide factors runs the cooperative bound with members of 128 columns (six per 768-wide factor);
    PSGDK_NLB_NARROW=0 (read at bind) keeps members of 256.  Both against the multi-launch route on the same inputs and draws, with the
    soak test's bounds (what may differ: the order of the fp32 row-sum atomics)."""
    dt, ulp = torch.bfloat16, 2.0 ** -7
    for env, cols in (("0", 256), (None, 128)):
        if env is None:
            monkeypatch.delenv("PSGDK_NLB_NARROW", raising=False)
        else:
            monkeypatch.setenv("PSGDK_NLB_NARROW", env)
        eng, _ = _engine(12, 768, dt)
        info = eng.info()
        assert info["nlb_coop"] == 1 and info["nlb_member_cols"] == cols, info
        F, dp = info["dense_factors"], info["max_dense_dim"]
        vsq_ref = torch.zeros(F, 4, 32, device=DEV)
        v_ref = torch.zeros(F, 2, 32, dp, device=DEV, dtype=dt)
        vsq, v = torch.zeros_like(vsq_ref), torch.zeros_like(v_ref)
        for chain in (0, 1):
            for it in range(20):
                _run_bound(eng, chain, 0, 7000 + it, vsq_ref, v_ref)
                _run_bound(eng, chain, 1, 7000 + it, vsq, v)
                rowmax = v_ref.float().abs().amax(dim=-1, keepdim=True)
                assert float(((v.float() - v_ref.float()).abs() / rowmax.clamp_min(1e-30)).amax()) <= 2 * ulp
                assert float(((vsq - vsq_ref).abs() / vsq_ref.abs().clamp_min(1e-30)).amax()) <= 1e-2
        assert eng.info()["nlb_fallbacks"] == 0
