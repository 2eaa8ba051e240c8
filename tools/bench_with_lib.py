#!/usr/bin/env python3
"""Runs bench.py against ANOTHER build of the library (an experiment compiled to its own file):  python tools/bench_with_lib.py <lib.so> [bench.py arguments]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
from psgd_torch_amd import _lib      # noqa: E402
_lib.LIB_PATH = lib
import bench                         # noqa: E402
sys.exit(bench.main())
