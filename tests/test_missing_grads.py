"""Parameters without a gradient on some steps (the reference skips them: wrapped_as_torch_optimizer_for_ddp.py:113-115).

The batched engine covers a fixed set of tensors; when the set of parameters with gradients changes, KWNS4 splits the
bucket into one single-tensor engine per parameter, carrying the state over.  Checked on CPU (TEST-ONLY OracleEngine for
the compute) against the oracle's restatement of the reference loop, which skips parameter by parameter: identical
parameters and preconditioners, per-parameter step counters that do not advance on skipped steps."""
import torch

from oracle import psgd_oracle as orc
from oracle_engine import OracleEngine

SHAPES = [(12, 8), (8,), (6, 6), (1, 5, 1), (10, 14)]
PATTERN = [set(), {2}, set(), {0, 4}, {0, 4}, set(), {1}]          # indices WITHOUT a gradient at each step


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(0.5 * torch.randn(s, generator=g)) for s in SHAPES]


def test_missing_gradients_follow_the_reference_loop():
    import psgd_torch_amd
    kw = dict(preconditioner_dtype=torch.float32, lr_params=1e-2, lr_preconditioner=0.3, momentum=0.9, weight_decay=0.01)
    pa = _params(3)
    opt = psgd_torch_amd.KWNS4(pa, engine_factory=OracleEngine, seed=5, **kw)

    pb = [p.detach().clone() for p in _params(3)]
    gate = torch.Generator().manual_seed(5)                          # KWNS4's host gate stream: group gate, then one draw per tensor
    eng = OracleEngine([()], "cpu")                                   # only for its (seed, offset, tensor id) -> generator hash
    steps = [0] * len(SHAPES)
    queue = []

    def uniform():
        return float(torch.rand([], generator=gate))

    def noise_for(G, kinds):
        i = queue.pop(0)
        # OracleEngine.update_precond: noise from hash(seed, offset = 2 t, tensor id = position in the group), gate from the host
        nz = orc.KronNoise.draw(G, kinds, eng._gen(5, 2 * (steps[i] - 1), i))
        nz.balance_u = 0.0 if uniform() < 0.01 else 1.0
        return nz

    ref = orc.KWNS4Oracle(pb, uniform=uniform, noise_for=noise_for, **kw)
    gg = torch.Generator().manual_seed(99)
    for missing in PATTERN:
        grads = [0.3 * torch.randn(s, generator=gg) for s in SHAPES]
        for i, p in enumerate(pa):
            p.grad = None if i in missing else grads[i].clone()
        opt.step()
        queue[:] = [i for i in range(len(SHAPES)) if i not in missing]
        for i in queue:
            steps[i] += 1
        ref.step([None if i in missing else grads[i] for i in range(len(SHAPES))])
    for i, (a, b) in enumerate(zip(pa, pb)):
        assert torch.equal(a.data, b), f"parameter {i} differs from the reference loop"
        assert opt.state[a]["step"] == steps[i] == ref.state[i]["step"]
        for qa, qb in zip(opt.state[a]["QL"][0], ref.state[i]["QL"][0]):
            assert torch.equal(qa, qb)


def test_checkpoint_round_trip_after_a_bucket_split():
    """state_dict()/load_state_dict() once the batched bucket has been split into per-parameter engines (the set of
    parameters with gradients changed): N steps + save + M steps == fresh optimizer, load, M steps -- bit for bit -- and the
    saved param_groups carry 'params' index lists like any torch optimizer's."""
    import copy
    import psgd_torch_amd
    kw = dict(preconditioner_dtype=torch.float32, lr_params=1e-2, lr_preconditioner=0.3, momentum=0.9, weight_decay=0.01)
    pa = _params(3)
    oa = psgd_torch_amd.KWNS4(pa, engine_factory=OracleEngine, seed=5, **kw)
    gg = torch.Generator().manual_seed(99)
    stream = [[0.3 * torch.randn(s, generator=gg) for s in SHAPES] for _ in range(len(PATTERN) + 3)]

    def run(opt, params, lo, hi):
        for t in range(lo, hi):
            missing = PATTERN[t] if t < len(PATTERN) else set()
            for i, p in enumerate(params):
                p.grad = None if i in missing else stream[t][i].clone()
            opt.step()
    run(oa, pa, 0, 4)                       # the split happens at step 1
    assert oa._split
    sd = copy.deepcopy(oa.state_dict())
    assert sd["param_groups"][0]["params"] == list(range(len(SHAPES)))
    snap = [p.detach().clone() for p in pa]
    run(oa, pa, 4, len(stream))
    pb = [torch.nn.Parameter(x.clone()) for x in snap]
    ob = psgd_torch_amd.KWNS4(pb, engine_factory=OracleEngine, seed=123, **kw)      # (the seed comes from the checkpoint)
    ob.load_state_dict(sd)
    run(ob, pb, 4, len(stream))
    for i, (a, b) in enumerate(zip(pa, pb)):
        assert torch.equal(a.data, b.data), f"parameter {i} differs after resume"
        assert oa.state[a]["step"] == ob.state[b]["step"]
        for qa, qb in zip(oa.state[a]["QL"][0], ob.state[b]["QL"][0]):
            assert torch.equal(qa, qb)


def test_resume_right_before_the_first_missing_gradient():
    """Checkpoint taken while the bucket is still batched, resumed at a step on which a parameter has no gradient: the batched
    bucket is rebuilt over the parameters it HAD (from the checkpoint), its state restored, and only then split -- bit for bit the
    uninterrupted run."""
    import copy
    import psgd_torch_amd
    kw = dict(preconditioner_dtype=torch.float32, lr_params=1e-2, lr_preconditioner=0.3, momentum=0.9, weight_decay=0.01)
    gg = torch.Generator().manual_seed(99)
    stream = [[0.3 * torch.randn(s, generator=gg) for s in SHAPES] for _ in range(6)]
    pattern = [set(), set(), {0, 3}, {3}, set(), {1}]

    def run(opt, params, lo, hi):
        for t in range(lo, hi):
            for i, p in enumerate(params):
                p.grad = None if i in pattern[t] else stream[t][i].clone()
            opt.step()
    pa = _params(4)
    oa = psgd_torch_amd.KWNS4(pa, engine_factory=OracleEngine, seed=5, **kw)
    run(oa, pa, 0, 2)
    assert not oa._split
    sd = copy.deepcopy(oa.state_dict())
    snap = [p.detach().clone() for p in pa]
    run(oa, pa, 2, 6)
    assert oa._split
    pb = [torch.nn.Parameter(x.clone()) for x in snap]
    ob = psgd_torch_amd.KWNS4(pb, engine_factory=OracleEngine, seed=77, **kw)
    ob.load_state_dict(sd)
    run(ob, pb, 2, 6)
    for i, (a, b) in enumerate(zip(pa, pb)):
        assert torch.equal(a.data, b.data), f"parameter {i} differs after resume"
        assert oa.state[a]["step"] == ob.state[b]["step"]
