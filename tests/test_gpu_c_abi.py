"""The drop-in boundary from plain C (tests/c_abi/abi_demo.c, compiled by __graft_entry__.build() with gcc: no Python, no torch in
the process): plan -> arenas from hipMalloc -> accumulate / update / precond_grad / apply_update on raw device pointers, with a
known-answer whitening check inside the program."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_demo_runs_and_whitens():
    # (the GPU box gets the prebuilt binary with the snapshot; build_c_abi_demo rebuilds it when it is missing or older than the header or
    #  the source -- a binary compiled against the previous PSGDK_VERSION refuses the library, which is its job)
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    exe = ge.build_c_abi_demo()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "abi_demo ok" in r.stdout, r.stdout
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libpsgdk.so" in ldd and "libtorch" not in ldd and "libpython" not in ldd and "libc10" not in ldd, ldd
