"""RCCL itself, as far as one GPU allows: a process group of ONE rank on the nccl (= RCCL) backend.  RCCL refuses two ranks on one device, so
the N > 1 tests of this suite travel over gloo; what they cannot show is that RCCL / torch's NCCL binding ACCEPT the calls the sharded path
issues on device memory -- the in-place all_gather_into_tensor whose send buffer is this rank's segment of the receive buffer, the grouped
point-to-point form, an all-reduce of one device float, the barrier.  Here those calls run for real (RCCL kernels / copies on the stream, world
size 1), driven by the product's own code: KWNS4 with the sharded step forced on, and the row-sharded LRA driver.  The results must equal the
unsharded optimizer's within the sharded-vs-replicated bounds of test_gpu_sharded.py."""
import os
import socket
import sys
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

SHAPES = [(96, 64), (64,), (64, 64), (1, 48, 1), (40, 72), (72,), (3, 4, 5), (), (96, 64), (48,)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _steps(opt, params, n, dev):
    g = torch.Generator().manual_seed(99)
    for _ in range(n):
        for p in params:
            p.grad = (0.3 * torch.randn(p.shape, generator=g)).to(dev)
        opt.step()
    torch.cuda.synchronize()


def _worker(rank, port, outdir, exchange):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        import psgd_torch_amd
        from psgd_torch_amd.kwns4 import KWNS4
        assert str(torch.distributed.get_backend()).lower() == "nccl"
        g = torch.Generator().manual_seed(7)
        params = [torch.nn.Parameter((0.5 * torch.randn(s, generator=g)).to(dev)) for s in SHAPES]
        opt = KWNS4(params, lr_params=1e-2, preconditioner_dtype=torch.bfloat16, shard_exchange=exchange)
        # a world of one is "not distributed" to the constructor: switch the sharded step on by hand, before the first bucket is built
        opt.shard_state, opt._shard_chunks = True, 1
        assert KWNS4._device_backend_is_rccl(params[0].data)
        _steps(opt, params, 4, dev)
        b = next(iter(opt._buckets.values()))
        assert b.flat is not None and b.flat.is_cuda          # the exchange buffer of the sharded step was built and gathered in place
        t = torch.ones(1, device=dev)
        torch.distributed.all_reduce(t)                       # the form _reduce_sum / _reduce_max issue
        torch.distributed.barrier()
        torch.cuda.synchronize()
        torch.save({"params": [p.data.cpu() for p in params], "t": float(t)}, os.path.join(outdir, "r0.pt"))
    finally:
        torch.distributed.destroy_process_group()


@pytest.mark.parametrize("exchange", ["all_gather", "p2p"])
def test_sharded_step_over_rccl_world_of_one(exchange):
    import psgd_torch_amd
    dev = "cuda:0"
    g = torch.Generator().manual_seed(7)
    ref = [torch.nn.Parameter((0.5 * torch.randn(s, generator=g)).to(dev)) for s in SHAPES]
    opt = psgd_torch_amd.KWNS4(ref, lr_params=1e-2, preconditioner_dtype=torch.bfloat16)
    _steps(opt, ref, 4, dev)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(_free_port(), d, exchange), nprocs=1, join=True)
        r = torch.load(os.path.join(d, "r0.pt"))
    assert r["t"] == 1.0
    for k, (a, c) in enumerate(zip(r["params"], ref)):
        err = float((a - c.data.cpu()).abs().max() / (c.data.abs().max().cpu() + 1e-12))
        assert err <= 2e-2, ("sharded step over RCCL (world 1) vs the unsharded optimizer", k, err)
