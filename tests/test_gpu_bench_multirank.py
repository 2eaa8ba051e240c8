"""bench.py's N > 1 code path with the REAL HIP engine: two ranks on one device over gloo (`--backend gloo --same-device`, a
testing-only combination: RCCL refuses two ranks on one GPU), sharded and replicated, launched exactly as the driver launches
it (torch.distributed.run, one JSON line from rank 0)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("mode", ["sharded", "replicated", "auto"])
def test_bench_two_ranks_one_device(mode):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--backend", "gloo", "--same-device", "--parallelism", mode, "--config", "lenet5", "--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["value"] > 0
    assert out["config"]["state_finite_after_timed_region"] is True
    if mode == "auto":       # the warm-up probe timed all three modes and the line says which one ran
        assert sorted(out["config"]["parallelism_probe_ms"]) == ["replicated", "sharded", "sharded, one exchange"], out["config"]
        assert all(v > 0 for v in out["config"]["parallelism_probe_ms"].values())
    else:
        assert ("sharding" in out["config"]["parallelism"]) == (mode == "sharded")
