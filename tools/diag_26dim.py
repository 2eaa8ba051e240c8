"""Where does the 26-dim case (2^26 elements, 26 dense 2 x 2 factors) spend its time on the GPU?  Phase by phase, synchronised."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import psgd_torch_amd as amd
from psgd_torch_amd import _lib as L
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 26
sq = (2,) * nd
def tick(msg, t0):
    torch.cuda.synchronize(); print(f"{msg}: {time.time() - t0:.2f} s", flush=True); return time.time()
t = time.time()
G = 0.5 * torch.randn(sq, device="cuda:0")
t = tick("randn", t)
QL, exprs = amd.init_kron(torch.zeros(sq, device="cuda:0"), Scale=0.7)
eng = exprs[0]
t = tick("init_kron (plan create + bind + init)", t)
eng.state_changed(); t = tick("state_changed", t)
eng.accumulate([G], keep_grad=True); t = tick("accumulate", t)
eng.update_precond(L.SRC_GRAD, 0.2, 0.9, 1e-6, seed=1, offset=0, noise=None, balance_mask=[False]); t = tick("update_precond (Philox noise)", t)
eng.precond_grad(L.SRC_GRAD); t = tick("precond_grad", t)
h = eng.read_precond_grad(0); t = tick("read_precond_grad", t)
print("finite:", bool(torch.isfinite(h).all()), flush=True)
