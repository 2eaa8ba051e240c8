#!/usr/bin/env python3
"""Prints the dispatch sequence of ONE optimizer step from a rocprofv3 rocpd database (kernel-trace): kernel, grid,
duration and gap to the previous dispatch -- the per-stage view behind DESIGN.md's GEMM table.
Usage: python tools/rocpd_sequence.py <results.db> [anchor-kernel-substring (default accumulate_kernel)] [which (default -2)]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    anchor = sys.argv[2] if len(sys.argv) > 2 else "accumulate_kernel"
    which = int(sys.argv[3]) if len(sys.argv) > 3 else -2
    cols = [r[1] for r in db.execute("pragma table_info(rocpd_kernel_dispatch)")]
    gx = "d.grid_size_x" if "grid_size_x" in cols else "0"
    wx = "d.workgroup_size_x" if "workgroup_size_x" in cols else "1"
    rows = db.execute(f"select s.kernel_name, d.start, d.end, {gx}, {wx} from rocpd_kernel_dispatch d "
                      "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
    idx = [i for i, r in enumerate(rows) if anchor in r[0]]
    lo, hi = idx[which], idx[which + 1]
    prev = None
    print("| # | kernel | workgroups | us | gap us |")
    print("|---:|---|---:|---:|---:|")
    tot = 0
    for k, (name, st, en, g, w) in enumerate(rows[lo:hi]):
        name = re.sub(r"\(.*\)$", "", re.sub(r"\s*\[clone .*\]$", "", name))[:60]
        gap = 0 if prev is None else (st - prev) / 1e3
        prev = en
        tot += en - st
        print(f"| {k} | `{name}` | {g // max(w, 1)} | {(en - st) / 1e3:.1f} | {gap:.1f} |")
    print(f"\nkernel time {tot / 1e6:.3f} ms, wall {(rows[hi][1] - rows[lo][1]) / 1e6:.3f} ms")


if __name__ == "__main__":
    main()
