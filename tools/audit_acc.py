"""ISA audit of the kernels whose accumulators are NAMED AGPRs that only inline asm touches (gemm_w4.hiph, gemm_w4p.hiph): the compiler
does not know those registers hold anything, so it must never use an AGPR itself (a value parked there lands in an accumulator), and
the kernels must not spill.  Compiles each kernel alone with -save-temps and checks every instruction outside the asm blocks.

    python tools/audit_acc.py        (exit code 1 and a listing if anything is found)
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = [("gemm_w4.hiph", "template __global__ void gemm_nt_w4_kernel<4>(const GemmProblem*, const GemmTile*, int);", "gemm_nt_w4_kernel"),
           ("gemm_w4p.hiph", "template __global__ void gemm_nt_w4p_kernel<0>(const GemmProblem*, const GemmTile*, int);", "gemm_nt_w4p_kernel")]
AGPR = re.compile(r"(?<![\w.])a(\[\d+:\d+\]|\d+)\b")


def audit(header, inst, name):
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "k.hip")
        open(src, "w").write(f'#include "{os.path.join(ROOT, "psgd_torch_amd", "csrc", header)}"\n{inst}\n')
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", os.path.join(d, "k.o"), "-save-temps"],
                           cwd=d, capture_output=True, text=True)
        if r.returncode:
            return [f"{header}: hipcc failed: {r.stderr[-400:]}"]
        asm = [f for f in os.listdir(d) if f.endswith("gfx950.s")][0]
        lines = open(os.path.join(d, asm)).read().splitlines()
    bad, in_kernel, in_asm, n_inst = [], False, False, 0
    for i, ln in enumerate(lines):
        if ln.startswith("_Z") and name in ln.split(":")[0] and ":" in ln:
            in_kernel = True
            continue
        if not in_kernel:
            continue
        t = ln.strip()
        if t.startswith("s_endpgm"):
            in_kernel = False
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        n_inst += 1
        code = t.split(";")[0]
        if not in_asm and (AGPR.search(code) or "accvgpr" in code):
            bad.append(f"{header}:{i + 1}: compiler touches an AGPR: {code.strip()}")
        if "scratch_" in code:
            bad.append(f"{header}:{i + 1}: scratch access: {code.strip()}")
    if n_inst < 1000:
        bad.append(f"{header}: kernel {name} not found in the ISA ({n_inst} instructions)")
    return bad


def main():
    bad = []
    for k in KERNELS:
        bad += audit(*k)
    for b in bad[:40]:
        print(b)
    print("accumulator audit:", "FAILED" if bad else "ok")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
