#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05_j
# correctness of the touch variant (w4 var 7) against the 128 x 128 tiling
python - <<'PY' > gpurun_out/r05_j/w4_touch_check.txt 2>&1
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from psgd_torch_amd import _lib
lib = _lib.lib(); dev = "cuda:0"; st = _lib.current_stream()
BIG, W4 = 1024, 1 << 26
bad = 0
torch.manual_seed(0)
for (M, N, K) in [(256, 256, 128), (256, 256, 192), (320, 768, 768), (2304, 768, 768), (50304, 768, 768), (4160, 832, 448), (131072, 768, 768)]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    rc_ = torch.zeros(M, N, device=dev, dtype=torch.bfloat16); rt_ = torch.zeros(N, M, device=dev, dtype=torch.bfloat16)
    _lib.check(lib.psgdk_test_gemm_nt(A.data_ptr(), B.data_ptr(), rc_.data_ptr(), None, 0, M, N, K, K, K, N, M, 0, st))
    _lib.check(lib.psgdk_test_gemm_nt(A.data_ptr(), B.data_ptr(), None, rt_.data_ptr(), 0, M, N, K, K, K, N, M, 0, st))
    for rep in range(3):
        c = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16); t = torch.full((N, M), float("nan"), device=dev, dtype=torch.bfloat16)
        _lib.check(lib.psgdk_test_gemm_nt(A.data_ptr(), B.data_ptr(), c.data_ptr(), None, 0, M, N, K, K, K, N, M, BIG | W4 | (7 << 27), st))
        _lib.check(lib.psgdk_test_gemm_nt(A.data_ptr(), B.data_ptr(), None, t.data_ptr(), 0, M, N, K, K, K, N, M, BIG | W4 | (7 << 27), st))
        if not (torch.equal(c, rc_) and torch.equal(t, rt_)):
            bad += 1; print("MISMATCH", M, N, K, rep, int((c != rc_).sum()), int((t != rt_).sum()))
    print(M, N, K, "done", flush=True)
print("W4 TOUCH CHECK", "FAILED" if bad else "OK")
PY
tail -3 gpurun_out/r05_j/w4_touch_check.txt
timeout 600 python tools/stage_bench.py small 20,22,21,23,40,41 > gpurun_out/r05_j/stage_bench_small.txt 2>&1; echo "rc $?"
grep -i "upd_a\|app_a" gpurun_out/r05_j/stage_bench_small.txt
timeout 600 python tools/stage_bench.py medium 20,22,21,23,40,41 > gpurun_out/r05_j/stage_bench_medium.txt 2>&1; echo "rc $?"
grep -i "upd_a\|app_a" gpurun_out/r05_j/stage_bench_medium.txt
