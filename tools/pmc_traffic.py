#!/usr/bin/env python3
"""HBM bytes per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd databases), per kernel.
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request on wide coalesced
streams -> read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 as is.
Usage: python tools/pmc_traffic.py <fetch.db> <write.db> > profiles/<name>.json"""
import json
import re
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select name, counter_value from pmc_events where counter_name = ?", (counter,)).fetchall()
    out = {}
    for name, v in rows:
        name = re.sub(r"\s*\[clone .*\]$", "", name)
        m = re.match(r"_Z\d+([A-Za-z0-9_]+?)I([tf])(?:Lb\d+E)?E", name)
        d = re.match(r"void (\w+)<(unsigned short|float)(?:, \w+)?>", name)
        if m:
            key = m.group(1) + "I" + m.group(2)
        elif d:
            key = d.group(1) + ("It" if d.group(2) == "unsigned short" else "If")
        else:
            key = name[:60]
        a = out.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += float(v)
    return out


def main():
    f = per_kernel(sys.argv[1], "FETCH_SIZE")
    w = per_kernel(sys.argv[2], "WRITE_SIZE")
    import hashlib, os
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "psgd_torch_amd", "libpsgdk.so")
    sha = hashlib.sha256(open(so, "rb").read()).hexdigest()[:16] if os.path.exists(so) else None
    res = {"library_sha256": sha,      # (bench.py compares it with the library it runs: roofline.traffic_stale)
           "note": "rocprofv3 --kernel-trace --pmc, separate passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --steps 3 --warmup 1`. "
                   "read bytes = 2 x FETCH_SIZE x 1024 (gfx950: 64 B counted per 128-B request), write bytes = WRITE_SIZE x 1024.",
           "kernels": {}}
    for k in sorted(set(f) | set(w)):
        nf, sf = f.get(k, [0, 0.0])
        nw, sw = w.get(k, [0, 0.0])
        n = max(nf, nw)
        if n == 0:
            continue
        fk, wk = (sf / nf if nf else 0.0), (sw / nw if nw else 0.0)
        res["kernels"][k] = {"dispatches_traced": n, "fetch_size_kb_per_launch": fk, "write_size_kb_per_launch": wk,
                             "hbm_bytes_per_launch_corrected": (2 * fk + wk) * 1024}
    g = [res["kernels"][k] for k in res["kernels"] if k.startswith("gemm_nt")]
    if g:
        tot = sum(x["hbm_bytes_per_launch_corrected"] * x["dispatches_traced"] for x in g)
        n = sum(x["dispatches_traced"] for x in g)
        res["gemm_all_tilings"] = {"dispatches_traced": n, "hbm_bytes_per_launch_corrected": tot / n}
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
