"""The RCCL code path of the sharded step on CPU: KWNS4's exchange takes different branches on RCCL (`torch.distributed.get_backend() == "nccl"`: device tensors
handed to the point-to-point requests, IN-PLACE all-gathers whose input aliases the output) than on gloo (host-staged requests, out-of-place gathers), and the
gloo tests never enter them.  torch's fake process group (every collective completes at once and moves nothing) with `get_backend` answering "nccl" runs exactly
those branches -- the aliasing contract asserted in `KWNS4._exchange`, the batched isend / irecv lists, the generator-driven order of the row-split exchange --
for one rank of an 8-rank job (what tools/rank_arithmetic.py does on the GPU).  The numbers are not a training state (the peers' segments stay zero); what is
checked is that the path runs, owns what the owner map says, keeps everything finite and asks the transport for the documented operations."""
import os
import subprocess
import sys
import textwrap

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

CHILD = textwrap.dedent("""
    import sys, json, collections
    sys.path.insert(0, {root!r}); sys.path.insert(0, {here!r})
    import torch, torch.distributed as dist
    from torch.testing._internal.distributed.fake_pg import FakeStore
    world, rank, chunks, exchange = {world}, {rank}, {chunks}, {exchange!r}
    dist.init_process_group(backend="fake", store=FakeStore(), rank=rank, world_size=world)
    dist.get_backend = lambda *a, **k: "nccl"
    calls = collections.Counter()
    for name in ("all_gather_into_tensor", "batch_isend_irecv", "all_reduce"):
        orig = getattr(dist, name)
        def wrap(*a, _o=orig, _n=name, **k):
            calls[_n] += 1
            if _n == "all_gather_into_tensor":      # RCCL path: the input is the rank's own segment of the output
                out, inp = a[0], a[1]
                assert inp.data_ptr() == out.data_ptr() + rank * inp.numel() * inp.element_size(), "all-gather input does not alias its segment"
            return _o(*a, **k)
        setattr(dist, name, wrap)
    import bench, psgd_torch_amd
    from oracle_engine import OracleEngine
    shapes = bench.gpt2_shapes(n_layer=2, n_embd=64, vocab=4160, block=128)
    g = torch.Generator().manual_seed(7)
    ps = [torch.nn.Parameter(0.5 * torch.randn(s, generator=g)) for s in shapes]
    opt = psgd_torch_amd.KWNS4(ps, preconditioner_dtype=torch.float32, engine_factory=OracleEngine, lr_params=1e-2, shard_state=True,
                               shard_chunks=chunks, shard_exchange=exchange)
    for _ in range(3):
        for p in ps:
            p.grad = 0.3 * torch.randn(p.shape, generator=g)
        opt.step()
    owned = sum(int(o == rank) for b in opt._buckets.values() for (_, _, _, o) in b.pieces)
    split = sum(len(b.blocks) for b in opt._buckets.values())
    print("RESULT " + json.dumps(dict(owned=owned, split=split, buckets=len(opt._buckets), finite=all(bool(torch.isfinite(p).all()) for p in ps),
                                      uneven=[bool(b.uneven) for b in opt._buckets.values()], calls=dict(calls))))
""")


@pytest.mark.parametrize("chunks,exchange", [(1, "all_gather"), (2, "all_gather"), (4, "p2p")])
def test_rccl_branches_of_the_sharded_step_run_under_the_fake_process_group(chunks, exchange):
    import json
    world, rank = 8, 3
    src = CHILD.format(root=ROOT, here=HERE, world=world, rank=rank, chunks=chunks, exchange=exchange)
    r = subprocess.run([sys.executable, "-c", src], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][0][7:])
    assert out["finite"] and out["split"] == 1 and out["owned"] >= 2, out          # the embedding is split; this rank owns its block and whole tensors
    assert out["buckets"] == chunks
    # one in-place all-gather of the row-split tensor's records per step, plus one per LEVEL chunk; ragged chunks (and every chunk in p2p mode) go point to point
    level = sum(1 for u in out["uneven"] if not u) if exchange == "all_gather" else 0
    assert out["calls"].get("all_gather_into_tensor", 0) == 3 * (1 + level), out
    assert out["calls"].get("batch_isend_irecv", 0) == 3 * (chunks - level), out
