#!/usr/bin/env python3
"""A/B helper: runs bench.py once per value of one environment variable (interleaved, repeated) and prints ms/step and the
grouped-GEMM rate.  Usage: python tools/env_sweep.py VAR v1,v2,... [repeats] [-- extra bench args]"""
import json
import os
import subprocess
import sys


def main():
    var, vals = sys.argv[1], sys.argv[2].split(",")
    rest = sys.argv[3:]
    reps = int(rest.pop(0)) if rest and rest[0].isdigit() else 2
    extra = rest[1:] if rest and rest[0] == "--" else rest
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for r in range(reps):
        for v in vals:
            env = dict(os.environ, **{var: v})
            out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "40", "--warmup", "10",
                                  "--no-apply-only"] + extra, env=env, capture_output=True, text=True).stdout
            d = json.loads(out.strip().splitlines()[-1])
            print(f"{var}={v}: {d['ms_per_step']:.4f} ms  gemm {d['roofline']['achieved']:.1f} {d['roofline']['unit']}"
                  f"  gemm_ms {d['roofline'].get('gemm_ms_per_step', 0):.4f}", flush=True)


if __name__ == "__main__":
    main()
