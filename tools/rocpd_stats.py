#!/usr/bin/env python3
"""Summarises a rocprofv3 rocpd database (kernel-trace) into a per-kernel stats table (what `--stats` prints for the
CSV format).  Usage: python tools/rocpd_stats.py <results.db> [--skip-first N] > profiles/<name>.md"""
import re
import sqlite3
import sys


def main():
    path = sys.argv[1]
    db = sqlite3.connect(path)
    rows = db.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d "
                      "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
    stats = {}
    for name, st, en in rows:
        name = re.sub(r"\s*\[clone .*\]$", "", name)
        name = re.sub(r"\(.*\)$", "", name)
        if len(name) > 90:
            name = name[:87] + "..."
        a = stats.setdefault(name, [0, 0, 1 << 62, 0])
        dur = en - st
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    total = sum(a[1] for a in stats.values())
    print(f"# rocprofv3 kernel-trace summary of `{path.split('/')[-1]}`\n")
    print(f"total kernel time {total/1e6:.3f} ms over {len(rows)} dispatches\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, a in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{name}` | {a[0]} | {a[1]/1e6:.3f} | {a[1]/a[0]/1e3:.1f} | {a[2]/1e3:.1f} | {a[3]/1e3:.1f} | {100*a[1]/total:.1f} |")


if __name__ == "__main__":
    main()
