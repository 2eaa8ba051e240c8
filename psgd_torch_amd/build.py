"""Builds libpsgdk.so (the HIP engine behind include/psgdk.h) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the built .so travels with the source
tree to the GPU box (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "psgdk.hip")
OUT = os.path.join(HERE, "libpsgdk.so")
# the measurement probes (matrix-core / HBM ceilings of the chip for bench.py's peak_measured) are a library of their own:
# the product library carries no probe kernels
PROBE_SRC = os.path.join(HERE, "csrc", "psgdk_probe.hip")
PROBE_OUT = os.path.join(HERE, "libpsgdk_probe.so")


def _sources():
    d = os.path.join(HERE, "csrc")
    return [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith((".hip", ".hiph"))] + [
        os.path.join(os.path.dirname(HERE), "include", "psgdk.h")]


def needs_build() -> bool:
    if not os.path.exists(OUT) or not os.path.exists(PROBE_OUT):
        return True
    t = min(os.path.getmtime(OUT), os.path.getmtime(PROBE_OUT))
    return any(os.path.getmtime(s) > t for s in _sources())


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    for src, out in ((SRC, OUT), (PROBE_SRC, PROBE_OUT)):
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-result", src, "-o", out]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
