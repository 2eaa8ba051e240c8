#!/usr/bin/env python3
"""Per-kernel SQ counter summary from one rocprofv3 --pmc pass (rocpd database): MFMA pipe utilisation and where the wave
cycles went.  Usage: python tools/pmc_sq.py <results.db> > profiles/<name>.json
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration in shader cycles x 1024 SIMDs); SQ_WAVE_CYCLES / SQ_WAIT_* /
SQ_ACTIVE_INST_ANY are quad-cycle counts that split the waves' lifetime (MI355X_MICROARCH.md, PMC section)."""
import json
import re
import sqlite3
import sys


def key_of(name):
    name = re.sub(r"\s*\[clone .*\]$", "", name)
    m = re.match(r"_Z\d+([A-Za-z0-9_]+?)I([tf])(?:Lb\d+E)?E", name)
    d = re.match(r"void (\w+)<(unsigned short|float)(?:, \w+)?>", name)
    if m:
        return m.group(1) + "I" + m.group(2)
    if d:
        return d.group(1) + ("It" if d.group(2) == "unsigned short" else "If")
    return name[:50]


def main():
    db = sqlite3.connect(sys.argv[1])
    clock_ghz = float(sys.argv[2]) if len(sys.argv) > 2 else 2.4
    rows = db.execute("select name, dispatch_id, start, end, counter_name, counter_value from pmc_events").fetchall()
    per = {}
    for name, did, st, en, cn, cv in rows:
        k = key_of(name)
        a = per.setdefault(k, {"disp": {}, "c": {}})
        a["disp"][did] = en - st
        a["c"][cn] = a["c"].get(cn, 0.0) + float(cv)
    out = {"note": f"sums over all traced dispatches of each kernel; shader clock assumed {clock_ghz} GHz for MfmaUtil", "kernels": {}}
    for k, a in per.items():
        c = a["c"]
        if "SQ_WAVE_CYCLES" not in c:
            continue
        dur_ns = sum(a["disp"].values())
        e = {"dispatches": len(a["disp"]), "total_us": dur_ns / 1e3, "counters": c}
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            e["MfmaUtil"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (dur_ns * clock_ghz * 1024)
        wc = c["SQ_WAVE_CYCLES"]
        for nm in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if nm in c and wc:
                e[nm + "_frac_of_wave_cycles"] = c[nm] / wc
        if c.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
        out["kernels"][k] = e
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
