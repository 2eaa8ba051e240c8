"""Registers, scratch and LDS of every kernel in libpsgdk.so, read from the code object's metadata (no GPU needed):

    python tools/kernel_resources.py [substring ...]

A kernel with a non-zero spill count or scratch size in a hot path is a finding; the CPU suite checks the GEMM kernels
(tests/test_abi_and_host.py::test_hot_kernels_do_not_spill)."""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_fingerprint import LLVM, SO, code_object  # noqa: E402


def resources(so: str = SO) -> dict:
    with tempfile.TemporaryDirectory() as tmp:
        co = code_object(so, tmp)
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    out = {}
    cur = {}
    for line in txt.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k in ("agpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "sgpr_count", "sgpr_spill_count", "vgpr_count",
                 "vgpr_spill_count", "max_flat_workgroup_size"):
            cur[k] = int(v) if v else 0
        elif k == "symbol":
            cur["symbol"] = v.strip("'")
        elif k == "wavefront_size":          # the last key of a kernel's record
            if "symbol" in cur:
                out[cur["symbol"]] = cur
            cur = {}
    return out


if __name__ == "__main__":
    res = resources()
    pats = sys.argv[1:]
    print(f"{'kernel':90s} vgpr agpr sgpr spillv spills scratch    lds")
    for sym, r in sorted(res.items()):
        if pats and not any(p in sym for p in pats):
            continue
        print(f"{sym[:90]:90s} {r.get('vgpr_count', -1):4d} {r.get('agpr_count', 0):4d} {r.get('sgpr_count', -1):4d} "
              f"{r.get('vgpr_spill_count', 0):6d} {r.get('sgpr_spill_count', 0):6d} {r.get('private_segment_fixed_size', 0):7d} "
              f"{r.get('group_segment_fixed_size', 0):6d}")
