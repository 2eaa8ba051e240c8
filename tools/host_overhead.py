"""Host-side cost of KWNS4.step() in isolation: the optimizer's own Python plus the engine's per-call argument work (tensor checks,
pointer tables), with every library call replaced by nothing.  Runs on CPU -- no GPU, no libpsgdk:

    python tools/host_overhead.py [gpt2-small|gpt2-medium|lenet5] [--profile] [--world N --rank R]

What it is for: on one GPU the host keeps ahead of a 1.7 ms step, but a rank of an 8-way sharded step has ~0.3 ms of kernels left
and the SAME Python in front of them; and LeNet5's whole step is 0.23 ms.  `--profile` prints the cProfile top list.
`--world N` builds the optimizer as rank R of N in sharded mode on a fake process group (no communication: the exchange is stubbed).
"""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from psgd_torch_amd import _lib as L                                     # noqa: E402
from psgd_torch_amd.engine import _check_tensors, _numel                  # noqa: E402
from psgd_torch_amd.kwns4 import KWNS4                                   # noqa: E402


def shapes_of(config):
    if config == "lenet5":
        return [(6, 26), (16, 151), (120, 257), (84, 121), (10, 85)]
    d, layers = (768, 12) if config == "gpt2-small" else (1024, 24)
    s = [(50304, d), (1024, d)]
    for _ in range(layers):
        s += [(d,), (d,), (3 * d, d), (3 * d,), (d, d), (d,), (d,), (d,), (4 * d, d), (4 * d,), (d, 4 * d), (d,)]
    return s + [(d,), (d,)]


class _NullFlat:
    def __init__(self, numels, offsets, device):
        self.numels = [int(x) for x in numels]
        self.device = torch.device(device)

    def apply(self, params, flat, lr, decoupled_wd):
        live = [(p, n) for p, n in zip(params, self.numels) if p is not None]
        _check_tensors("params", [p for p, _ in live], [n for _, n in live], self.device)
        self._keep = (L.ptr_array(params), list(params), flat)


class NullEngine:
    """KronEngine's per-call host work without the library: the same checks and pointer tables, no launches."""
    FlatApply = _NullFlat

    def __init__(self, shapes, device, precond_dtype=torch.bfloat16, max_size=float("inf"), max_skew=1.0, use_momentum=True,
                 init_scale=1.0, tensor_ids=None, geometry="Q0.5EQ1.5"):
        self.device = torch.device(device)
        self.shapes = [tuple(s) for s in shapes]
        self.n = len(self.shapes)
        self.numels = [_numel(s) for s in self.shapes]
        self.dtype = precond_dtype
        self.ema = [torch.empty(0) for _ in shapes]
        self.state_arena = torch.empty(0)

    def QL(self, k):
        return [[], []]

    def accumulate(self, grads, params=None, coupled_wd=0.0, beta=0.0, keep_grad=False, damp=None):
        _check_tensors("grads", grads, self.numels, self.device)
        if params is not None:
            _check_tensors("params", params, self.numels, self.device)
        self._keep = [L.ptr_array(grads), L.ptr_array(params) if params is not None else None, list(grads)]
        if damp is not None:
            self._keep.append(L.Damp(int(damp["source"]), float(damp["damping"]), None, int(damp["seed"]), int(damp["offset"])))

    def update_precond(self, source, lr, betaL, damping, seed=0, offset=0, noise=None, balance_mask=None):
        import ctypes as C
        if balance_mask is not None:
            self._bm = (C.c_uint8 * self.n)(*[1 if b else 0 for b in balance_mask])

    def precond_grad(self, source):
        pass

    def apply_update(self, params, lr, decoupled_wd, max_avg_amp, max_elem_amp):
        _check_tensors("params", params, self.numels, self.device)
        self._keep_p = [L.ptr_array(params), list(params)]

    def export_precond_grad(self, outs, clip=True, max_avg_amp=2.0, max_elem_amp=10.0):
        _check_tensors("outs", outs, self.numels, self.device)
        self._keep_o = [L.ptr_array(outs), list(outs)]

    def state_changed(self):
        pass

    def info(self):
        return {"nlb_coop": 0}


class _Done:
    def wait(self):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", nargs="?", default="gpt2-small")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--world", type=int, default=1)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--chunks", type=int, default=4)
    a = ap.parse_args()
    torch.set_num_threads(1)
    # storage is irrelevant here: every tensor of a shape shares one buffer (1.4 GB of medium-size parameters otherwise)
    params = [torch.nn.Parameter(torch.empty(s)) for s in shapes_of(a.config)]
    for p in params:
        p.grad = torch.empty_like(p)
    kw = {}
    if a.world > 1:
        kw = dict(shard_state=True, shard_chunks=a.chunks)
    opt = KWNS4(params, engine_factory=NullEngine, preconditioner_dtype=torch.float32 if a.config == "lenet5" else torch.bfloat16, **kw)
    if a.world > 1:      # what __init__ would have read from an initialised process group
        opt.is_distributed, opt.world, opt.rank = True, a.world, a.rank
        opt.shard_state, opt._shard_chunks = True, max(1, a.chunks)
        opt._exchange = lambda b: _Done()
    for _ in range(20):
        opt.step()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        opt.step()
    dt = (time.perf_counter() - t0) / a.steps
    print(f"{a.config}: {len(params)} tensors, world {a.world}: {dt * 1e3:.3f} ms of host Python per step "
          f"({dt / len(params) * 1e6:.2f} us per tensor)")
    if a.profile:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(a.steps):
            opt.step()
        pr.disable()
        st = pstats.Stats(pr)
        st.sort_stats("tottime").print_stats(18)


if __name__ == "__main__":
    main()
