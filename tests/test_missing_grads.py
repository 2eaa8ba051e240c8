"""Parameters without a gradient on some steps (the reference skips them: wrapped_as_torch_optimizer_for_ddp.py:113-115).

The batched engine covers a fixed set of tensors; when the set of parameters with gradients changes, KWNS4 splits the
bucket into one single-tensor engine per parameter, carrying the state over.  Checked on CPU (TEST-ONLY OracleEngine for
the compute) against the oracle's restatement of the reference loop, which skips parameter by parameter: identical
parameters and preconditioners, per-parameter step counters that do not advance on skipped steps."""
import pytest
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


@pytest.mark.parametrize("dQ", ["QUAD", "EQ", "QEQ", "QEP", "QUAD4P"])
def test_geometry_switch_and_strided_parameters_follow_the_reference_loop(dQ):
    """KWNS4(dQ=...) -- the reference's three-line switch (wrapped_as_torch_optimizer_for_ddp.py:84-86) as a keyword -- against the
    oracle's restatement of the loop with the same switch, bit for bit on CPU (TEST-ONLY OracleEngine for the compute).  Two of the
    parameters are NOT contiguous (a channels_last 4-D weight and a transposed view): the reference's `p.subtract_(h.view_as(p))`
    (..._ddp.py:157) is stride-safe, the engine takes packed shadows and KWNS4 copies them back."""
    import psgd_torch_amd
    kw = dict(preconditioner_dtype=torch.float32, lr_params=1e-2, lr_preconditioner=0.3, momentum=0.9, weight_decay=0.01,
              preconditioner_init_scale=0.8)
    g0 = torch.Generator().manual_seed(11)
    shapes = [(6, 4, 3, 3), (8, 12), (5,), (7, 7)]
    base = [0.5 * torch.randn(s, generator=g0) for s in shapes]
    pa = [torch.nn.Parameter(base[0].clone().contiguous(memory_format=torch.channels_last)),
          torch.nn.Parameter(base[1].t().contiguous().t()),                # logical (8, 12), stored column-major
          torch.nn.Parameter(base[2].clone()), torch.nn.Parameter(base[3].clone())]
    assert not pa[0].is_contiguous() and not pa[1].is_contiguous()
    opt = psgd_torch_amd.KWNS4(pa, engine_factory=OracleEngine, seed=5, dQ=dQ, **kw)
    assert opt.dQ == dQ
    pb = [b.clone() for b in base]
    gate = torch.Generator().manual_seed(5)
    eng = OracleEngine([()], "cpu")
    step = {"t": 0}
    queue = []

    def uniform():
        return float(torch.rand([], generator=gate))

    # KWNS4 draws the group gate, then one balancing gate per tensor BEFORE the batched update call (for every geometry, so that
    # the gate stream does not depend on dQ); the oracle loop asks per tensor, in order
    gates = []

    def noise_for(G, kinds):
        i = queue.pop(0)
        nz = orc.KronNoise.draw(G, kinds, eng._gen(5, 2 * step["t"], i))
        nz.balance_u = 0.0 if gates[i] < 0.01 else 1.0
        return nz

    class Ref(orc.KWNS4Oracle):
        pass
    ref = Ref(pb, uniform=uniform, noise_for=noise_for, dQ=dQ, **kw)
    gg = torch.Generator().manual_seed(99)
    for t in range(4):
        grads = [0.3 * torch.randn(s, generator=gg) for s in shapes]
        for p, g in zip(pa, grads):
            p.grad = g.clone()
        opt.step()
        step["t"] = t
        queue[:] = list(range(len(shapes)))
        # replay the host gate stream in KWNS4's order: group gate first (consumed inside ref.step via uniform()), then the per-tensor gates
        state = gate.get_state()
        uniform()                                                    # the group gate
        gates[:] = [uniform() for _ in shapes]
        after = gate.get_state()
        gate.set_state(state)
        ref.step(grads)                                              # draws the group gate again from the same position
        gate.set_state(after)
    for i, (a, b) in enumerate(zip(pa, pb)):
        assert a.data.stride() == pa[i].stride()
        assert torch.equal(a.data, b), f"parameter {i} differs from the reference loop ({dQ})"
        for qa, qb in zip(opt.state[a]["QL"][0], ref.state[i]["QL"][0]):
            assert torch.equal(qa, qb)
    sd = opt.state_dict()
    assert sd["dQ"] == dQ
    other = psgd_torch_amd.KWNS4([torch.nn.Parameter(b.clone()) for b in base], engine_factory=OracleEngine, seed=5, **kw)
    with pytest.raises(ValueError):
        other.load_state_dict(sd)


def test_checkpoint_does_not_depend_on_the_device_index():
    """The usual DDP flow: rank 0 (cuda:0) saves, every rank loads into the bucket of ITS device.  Checkpoint keys carry the device
    TYPE only when the optimizer lives on one device; files written with a device index (version-2 keys of round 2, or another
    rank's) are matched as well.  A checkpoint with entries that match nothing is reported."""
    import psgd_torch_amd
    kw = dict(preconditioner_dtype=torch.float32, lr_params=1e-2, lr_preconditioner=0.3, momentum=0.9)
    pa = _params(3)
    opt = psgd_torch_amd.KWNS4(pa, engine_factory=OracleEngine, seed=5, **kw)
    gg = torch.Generator().manual_seed(99)
    for _ in range(3):
        for p in pa:
            p.grad = 0.3 * torch.randn(p.shape, generator=gg)
        opt.step()
    sd = opt.state_dict()
    assert all("|cpu" in k and "cpu:" not in k for k in sd["buckets"])
    # what rank 0 on cuda:0 would have written in round 2 / what another rank's file looks like
    sd["buckets"] = {k.replace("|cpu", "|cuda:3"): v for k, v in sd["buckets"].items()}
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    opt2 = psgd_torch_amd.KWNS4(pb, engine_factory=OracleEngine, seed=5, **kw)
    opt2.load_state_dict(sd)
    for p, q in zip(pa, pb):
        g = 0.3 * torch.randn(p.shape, generator=gg)
        p.grad, q.grad = g.clone(), g.clone()
    opt.step(); opt2.step()
    assert not opt2._pending_restore
    for p, q in zip(pa, pb):
        assert torch.equal(p.data, q.data)
        for a, b in zip(opt.state[p]["QL"][0], opt2.state[q]["QL"][0]):
            assert torch.equal(a, b)


def test_batched_gate_draws_are_the_scalar_stream():
    """KWNS4._uniforms(n) draws the n per-tensor balancing gates of one update call with ONE generator call; it must be the stream
    n scalar draws give (values and final generator state), so that checkpoints, ranks and the recorded-draw tests of earlier rounds
    see the same gates -- and a replaced `_uniform` (how the golden tests replay the reference's draws) must still be honoured."""
    import psgd_torch_amd
    p = [torch.nn.Parameter(torch.zeros(3, 2))]
    for n in (1, 5, 16, 17, 148, 292):
        a = psgd_torch_amd.KWNS4(p, seed=7, engine_factory=OracleEngine)
        b = psgd_torch_amd.KWNS4(p, seed=7, engine_factory=OracleEngine)
        one_by_one = [a._uniform() for _ in range(n)]
        assert b._uniforms(n) == one_by_one
        assert torch.equal(a._gate_gen.get_state(), b._gate_gen.get_state())
        assert a._uniform() == b._uniform()
    c = psgd_torch_amd.KWNS4(p, seed=7, engine_factory=OracleEngine)
    seq = iter([0.25, 0.5, 0.75])
    c._uniform = lambda: next(seq)
    assert c._uniforms(3) == [0.25, 0.5, 0.75]
