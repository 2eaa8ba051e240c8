"""The parameter update fused into the epilogue of the apply's last product (psgdk_precond_grad_apply; -m gpu).

Reference lines: wrapped_as_torch_optimizer_for_ddp.py:150-157 (h = precond_grad; RMS clip; element clamp; p -= lr h) and :117-120
(decoupled weight decay).  The two-call route (psgdk_precond_grad + psgdk_apply_update) is what every golden test of rounds 1-5
pinned to the reference; here the fused route is compared with it on the same inputs:

  * no clip engaged: the parameters must be BIT-IDENTICAL (same operations, same roundings; the product's accumulators are the
    same), on both GEMM tilings, for tensors held as they are, held transposed, with two dense factors, with 1-D / N-D / odd-row
    tensors (which the streaming pass still updates inside the same call), with and without decoupled weight decay;
  * clip engaged on some tensors (speculative update + clip_fix_kernel): within one fp32 rounding of the parameter per step
    (p' + lr c1 - lr c2 instead of p keep - lr c2), tensors whose clip did not engage still bit-identical;
  * the golden KWNS4 trajectories (recorded from the reference) through the fused route: same bounds as the two-call route;
  * the GPT-2-small parameter list at full size (what bench.py times): bit-identical after two steps;
  * h is consumed: reading it afterwards is refused.
Also here (round 6): the fp32 instantiation of the 256 x 256 kernel sums ||h||^2 over all eight waves, and the momentum
pass's general path equals its compile-time-resolved path bit for bit (psgdk_test_ew_mode)."""
import pytest
import torch

import psgd_torch_amd
from psgd_torch_amd import _lib as L

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# (768, 256): [diag, dense] held as it is; (64, 512): dense dim first -> held transposed; (128, 128): two dense factors;
# (300,), (): diagonal only; (6, 5, 3, 3): N-D; (192, 66): row length not a multiple of 4 -> not fusable; (1, 320, 1): squeezes to 1-D
# (2048, 384), (384, 1536): enough 128 x 128 tiles that the apply's products leave the K-split kernel of small plans (which takes the
# two-call route whole) for the 128 x 128 kernel
SHAPES = [(768, 256), (64, 512), (128, 128), (300,), (), (6, 5, 3, 3), (192, 66), (1, 320, 1), (256, 192), (512, 64), (2048, 384), (384, 1536)]
N_FUSABLE = 7          # the 2-D tensors with a dense factor and a row length that is a multiple of 4


def _engine_with_fitted_factors(shapes, pd, seed=0):
    """A KronEngine whose factors have moved away from the identity (three updates on structured-ish gradients) and whose momentum holds
    one more gradient: the state both routes of the apply start from."""
    eng = psgd_torch_amd.KronEngine(shapes, DEV, precond_dtype=pd)
    gen = torch.Generator().manual_seed(seed)
    for k in range(3):
        gs = [(0.3 * torch.randn(s, generator=gen) * torch.linspace(0.2, 2.0, s[-1] if len(s) else 1)).reshape(s).to(DEV) for s in shapes]
        eng.accumulate(gs, beta=0.5, damp=dict(source=L.SRC_EMA, damping=1e-6, seed=7, offset=k))
        eng.update_precond(L.SRC_EMA, 0.3, 0.9, 1e-6, seed=7, offset=k)
    gs = [(0.3 * torch.randn(s, generator=gen)).to(DEV) for s in shapes]
    eng.accumulate(gs, beta=0.9)
    p0 = [(0.1 * torch.randn(s, generator=gen)).to(DEV) for s in shapes]
    torch.cuda.synchronize()
    return eng, p0


def _both_routes(eng, p0, lr, wd, max_avg, max_elem):
    """The SAME engine state through the fused call and through its two-call route (psgdk_test_fuse_mode): everything upstream of the
    parameter update -- Q, the momentum, hence h -- is identical by construction (two optimizer runs are not: the traces and row sums of
    the preconditioner update are accumulated with unordered fp32 atomics, which fp32 factors keep in their last bits)."""
    out = []
    for fuse in (True, False):
        ps = [p.clone() for p in p0]
        eng.fuse_update(fuse)
        eng.precond_grad_apply(L.SRC_EMA, ps, lr, wd, max_avg, max_elem)
        fused = eng.info()["update_fused"]
        torch.cuda.synchronize()
        out.append((ps, fused))
    eng.fuse_update(True)
    return out


@pytest.mark.parametrize("pd", [torch.bfloat16, torch.float32], ids=["bf16", "fp32"])
@pytest.mark.parametrize("big", [False, True], ids=["tile128", "tile256"])
@pytest.mark.parametrize("wd", [0.0, 0.05], ids=["nowd", "wd"])
def test_fused_update_equals_two_call_route_bitwise(pd, big, wd, monkeypatch):
    if big:
        monkeypatch.setenv("PSGDK_BIG_MIN_TILES", "1")          # read when a plan is bound: the 256 x 256 kernel wherever it can run
    eng, p0 = _engine_with_fitted_factors(SHAPES, pd)
    (a, fa), (b, fb) = _both_routes(eng, p0, 1e-2, wd, 1e6, 1e7)      # (the RMS clip cannot engage)
    # the fused epilogue really ran, on every tensor that allows it (tile128: the lone (128, 128) tensor's second product is a one-tile
    # launch on the K-split kernel, which keeps the two-pass update), and the other call really took the two-call route
    assert fa == (N_FUSABLE if big else N_FUSABLE - 1) and fb == 0, (fa, fb)
    eng.fuse_update(True)
    ps = [p.clone() for p in p0]
    eng.precond_grad_apply(L.SRC_EMA, ps, 1e-2, wd, 1e6, 1e7)
    with pytest.raises(L.PsgdkError):                            # h was consumed by the fused call
        eng.read_precond_grad(0)
    for s, x, y, z in zip(SHAPES, a, b, p0):
        assert torch.isfinite(x).all() and not torch.equal(x, z)
        assert torch.equal(x, y), f"{s}: fused update differs from the two-call route by {float((x - y).abs().max())}"


@pytest.mark.parametrize("pd", [torch.bfloat16, torch.float32], ids=["bf16", "fp32"])
@pytest.mark.parametrize("big", [False, True], ids=["tile128", "tile256"])
def test_fused_update_with_the_rms_clip_engaged(pd, big, monkeypatch):
    """max_avg below the RMS of some tensors' h and above that of others: the clipped ones go through the speculative update +
    clip_fix_kernel and may differ from the two-call route by one fp32 rounding of p; the others stay bit-identical."""
    if big:
        monkeypatch.setenv("PSGDK_BIG_MIN_TILES", "1")
    shapes = [(768, 256), (64, 512), (128, 128), (256, 192), (2048, 384), (384, 1536)]
    eng, p0 = _engine_with_fitted_factors(shapes, pd, seed=3)
    # RMS(h) per tensor from the two-call route's own h
    eng.precond_grad(L.SRC_EMA)
    rms = [float(eng.read_precond_grad(k).float().pow(2).mean().sqrt()) for k in range(len(shapes))]
    max_avg = sorted(rms)[len(rms) // 2] * 0.999          # the upper half clips
    clipped = [r > max_avg for r in rms]
    assert any(clipped) and not all(clipped), rms
    lr = 1e-2
    (a, _), (b, _) = _both_routes(eng, p0, lr, 0.01, max_avg, 2.5 * max_avg)
    for k, (x, y) in enumerate(zip(a, b)):
        if not clipped[k]:
            assert torch.equal(x, y), f"tensor {k} (no clip) must be bit-identical"
        else:
            # one fp32 rounding of p (|p| < 0.6: ulp <= 6e-8)
            assert float((x - y).abs().max()) <= 6e-8, (k, float((x - y).abs().max()))
            # ... and the clip did engage: no element moved further than lr * max_elem (+ the decay)
            assert float((x - p0[k]).abs().max()) <= lr * 2.5 * max_avg * 1.0001 + 0.01 * lr * 0.6 + 1e-7


def _kwns4_run(shapes, fuse, steps, pd, seed=5, dev_gen=True, **kw):
    gen = torch.Generator(device=DEV).manual_seed(seed)
    params = [torch.nn.Parameter(0.02 * torch.randn(*s, device=DEV, generator=gen)) for s in shapes]
    opt = psgd_torch_amd.KWNS4(params, preconditioner_dtype=pd, **kw)
    opt._fuse_update = fuse
    for _ in range(steps):
        for p in params:
            p.grad = 0.01 * torch.randn(p.shape, device=DEV, generator=gen)
        opt.step()
    torch.cuda.synchronize()
    fused = sum(b.engine.info()["update_fused"] for b in opt._buckets.values())
    return [p.detach().clone() for p in params], fused


def test_fused_update_on_the_gpt2_small_plan():
    """bench.py's plan through KWNS4.step(): 148 tensors, the persistent 256 x 256 kernel on the two full-size products (wte's 196.5 row
    tiles: the last one takes the edge form), the (768, 3072) c_proj tensors held transposed.  Two optimizer runs (bf16 factors round the
    update's unordered fp32 atomics away almost always, but not provably: the bound is one bf16 ulp of h times lr, not zero)."""
    from test_gpu_production_path import gpt2_shapes
    shapes = gpt2_shapes()
    a, fa = _kwns4_run(shapes, True, 2, torch.bfloat16)
    torch.cuda.empty_cache()
    b, fb = _kwns4_run(shapes, False, 2, torch.bfloat16)
    assert fa == 50 and fb == 0, (fa, fb)          # wte, wpe and the 48 matrices of the blocks
    same = 0
    for s, x, y in zip(shapes, a, b):
        assert float((x - y).abs().max()) <= 2e-4 * 10.0 * 2 ** -7, (s, float((x - y).abs().max()))      # lr x clamp x one bf16 ulp
        same += int(torch.equal(x, y))
    assert same >= len(shapes) - 2, same


def test_fp32_big_tiling_sums_all_waves():
    """in the fp32 instantiation of the 256 x 256 kernel the per-tile reduction of sum h^2 wrote six of its eight slots
    past the LDS array (dropped writes): ||h||^2 came out too small and the RMS clip under-clipped."""
    import os
    os.environ["PSGDK_BIG_MIN_TILES"] = "1"
    try:
        eng = psgd_torch_amd.KronEngine([(512, 256)], DEV, precond_dtype=torch.float32)
    finally:
        del os.environ["PSGDK_BIG_MIN_TILES"]
    gen = torch.Generator().manual_seed(0)
    g = torch.randn(512, 256, generator=gen).to(DEV)
    eng.accumulate([g], beta=0.0)
    eng.precond_grad(L.SRC_EMA)
    h = eng.read_precond_grad(0)
    torch.cuda.synchronize()
    want = float((h.double() ** 2).sum())
    got = float(eng.hsumsq[0])
    assert abs(got - want) <= 1e-4 * want, (got, want)


def test_momentum_pass_general_path_equals_fast_path_bitwise():
    """psgdk_test_ew_mode(2) sends every tile of accumulate_kernel through the general path; the interior-tile path must give the same
    bits (EMA and damped input), which the explicit fmaf in both now pins by construction."""
    shapes = [(1024, 768), (768, 3072)]
    outs = []
    for mode in (0, 2):
        eng = psgd_torch_amd.KronEngine(shapes, DEV, precond_dtype=torch.bfloat16)
        L.check(L.lib().psgdk_test_ew_mode(eng._plan, mode), "ew_mode")
        gen = torch.Generator().manual_seed(1)
        for k in range(2):
            gs = [torch.randn(s, generator=gen).to(DEV) for s in shapes]
            eng.accumulate(gs, beta=0.9, damp=dict(source=L.SRC_EMA, damping=1e-3, seed=11, offset=k))
        torch.cuda.synchronize()
        outs.append((eng.state_arena.clone(), eng.work_arena.clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])
