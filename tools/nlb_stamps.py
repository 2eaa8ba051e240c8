"""Where one cooperative norm-bound launch spends its time: per-phase wall-clock stamps from the instrumented instantiation of
nlb_coop_kernel (psgdk_test_nlb_stamps, include/psgdk_test.h).  GPU only:

    python tools/nlb_stamps.py [--width 768 --factors 62 --dtype bf16] [--reps 5]

Prints, per chain, the median / max over workgroups of every phase's duration (us), the spread of the workgroups' START times (the
launch ramp: a member cannot finish an exchange before its slowest sibling has started), how long members wait for their siblings in each
exchange, and which XCC the members of a factor ran on.  The instrumented kernel fences each phase (s_waitcnt + barrier) so the phases do not
overlap the way the production kernel lets them: the SUM is an upper bound of the production launch, the split is what it is for.
"""
import argparse
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch                                                     # noqa: E402

NAMES = {0: "entry", 1: "job, descriptor, scalars, arg-max row", 2: "start block (A[j], Philox, LDS)", 3: "rows of A -> registers",
         0x60: "scalars / exit"}
for p in range(4):
    NAMES[0x10 + p] = f"product {p}: MFMAs"
    NAMES[0x20 + p] = f"product {p}: scale, round, stage, issue stores"
    NAMES[0x30 + p] = f"product {p}: stores acknowledged + barrier"
    NAMES[0x40 + p] = f"product {p}: announce + wait for siblings"
    NAMES[0x50 + p] = f"product {p}: block back to LDS (DMA) + barrier"


def parse_blocks(words, n_blocks):
    """32 words per workgroup -> [{f, m, S, xcc, stamps: [(clock, code), ...]}], idle blocks of the padded job table dropped."""
    blocks = []
    for b in range(n_blocks):
        w = words[b * 32:(b + 1) * 32]
        meta = w[31]
        stamps = [(x >> 8, x & 0xff) for x in w[:30] if x]
        if len(stamps) < 3:
            continue
        blocks.append(dict(f=meta >> 32, m=(meta >> 8) & 0xffffff, S=meta & 0xff, xcc=w[30] & 0xf, stamps=stamps))
    return blocks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=768)
    ap.add_argument("--factors", type=int, default=62)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    from test_gpu_nlb import _engine
    from psgd_torch_amd import _lib as L
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    eng, _ = _engine(a.factors, a.width, dt)
    if not eng.info()["nlb_coop"]:
        raise SystemExit("this plan does not use the cooperative launch")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    max_blocks = 2048
    buf = (C.c_uint64 * (max_blocks * 32))()
    nb = C.c_int()
    for chain in (0, 1):
        per_phase, starts_spread, totals, waits = {}, [], [], {}
        xcc_mixed = 0
        for rep in range(a.reps + 1):
            L.check(eng.lib.psgdk_test_nlb_stamps(eng._plan, chain, 100 + rep, 0, buf, max_blocks, C.byref(nb), st), "test_nlb_stamps")
            if rep == 0:
                continue                                          # warm-up (code fetch)
            blocks = parse_blocks(buf, nb.value)
            t0 = min(b["stamps"][0][0] for b in blocks)
            starts_spread.append((max(b["stamps"][0][0] for b in blocks) - t0) / 100.0)
            totals.append((max(b["stamps"][-1][0] for b in blocks) - t0) / 100.0)
            for b in blocks:
                prev = b["stamps"][0][0]
                for t, code in b["stamps"][1:]:
                    per_phase.setdefault(code, []).append((t - prev) / 100.0)
                    prev = t
            byf = {}
            for b in blocks:
                byf.setdefault(b["f"], []).append(b)
            for f, ms in byf.items():
                if len({m["xcc"] for m in ms}) > 1:
                    xcc_mixed += 1
                # how much later than its own "stores acknowledged" stamp a member sees the last sibling: pure waiting
                for p in range(4):
                    acks = [t for m in ms for t, c in m["stamps"] if c == 0x30 + p]
                    if len(acks) > 1:
                        waits.setdefault(p, []).append((max(acks) - min(acks)) / 100.0)
        print(f"\n=== chain {chain} ({'spd: A = term1' if chain == 0 else 'skh: A = R'}), {a.factors} x {a.width} {a.dtype}, {nb.value} workgroups, "
              f"{a.reps} launches ===")
        print(f"launch total (first entry -> last exit): median {statistics.median(totals):.1f} us; start-time spread over workgroups: "
              f"median {statistics.median(starts_spread):.1f} us; factors with members on different XCCs: {xcc_mixed} of "
              f"{a.reps * a.factors}")
        print(f"{'phase':58s} {'median us':>10s} {'p90 us':>8s} {'max us':>8s}")
        tot = 0.0
        for code in sorted(per_phase):
            v = sorted(per_phase[code])
            med = statistics.median(v)
            tot += med
            print(f"{NAMES.get(code, hex(code)):58s} {med:10.2f} {v[int(0.9 * (len(v) - 1))]:8.2f} {v[-1]:8.2f}")
        print(f"{'sum of medians':58s} {tot:10.2f}")
        for p in sorted(waits):
            v = sorted(waits[p])
            print(f"  exchange {p}: spread of the siblings' 'stores acknowledged' times: median {statistics.median(v):.2f} us, max {v[-1]:.2f} us")


if __name__ == "__main__":
    main()
