"""Per-kernel fingerprint of the machine code in libpsgdk.so: a way to change the source WITHOUT a GPU at hand and know which
kernels' code actually changed.

    python tools/isa_fingerprint.py save  [file]     # fingerprints of the current build -> file (default gpurun_out/isa_baseline.json)
    python tools/isa_fingerprint.py check [file]     # rebuild-free comparison of the current .so against the saved fingerprints

A kernel's fingerprint is the SHA-1 of its disassembly (llvm-objdump of the gfx950 code object inside the fat binary) with addresses,
branch targets and encodings stripped.
`check` lists kernels that changed, disappeared or are new; exit code 1 if a kernel that existed before has different code.
Used in round 3 after the GPU budget was spent: experiments were added as NEW template instantiations behind environment switches, and
this check proves that every kernel of the validated build is byte-for-byte what the GPU tests ran.
"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "psgd_torch_amd", "libpsgdk.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(so: str, tmp: str) -> str:
    """The gfx950 ELF embedded in the host library's .hip_fatbin section."""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", so, fat], check=True)
    out = os.path.join(tmp, "gfx950.co")
    res = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                          "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={out}"], capture_output=True, text=True)
    if res.returncode != 0 or not os.path.exists(out) or os.path.getsize(out) == 0:
        raise SystemExit("could not unbundle the gfx950 code object:\n" + res.stderr)
    return out


def fingerprints(so: str = SO) -> dict:
    with tempfile.TemporaryDirectory() as tmp:
        co = code_object(so, tmp)
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "--no-leading-addr", co],
                             capture_output=True, text=True, check=True).stdout
    out, cur, body = {}, None, []

    def flush():
        if cur is not None:
            out[cur] = {"code": hashlib.sha1("\n".join(body).encode()).hexdigest(), "insns": len(body)}

    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]* ?<([^>]+)>:$", line.strip())
        if m:
            flush()
            cur, body = m.group(1), []
            continue
        s = line.strip()
        if not s or cur is None:
            continue
        s = re.sub(r"//.*$", "", s).strip()                    # address / encoding comments
        s = re.sub(r"<[^>]+>", "<L>", s)                        # symbolic branch targets (offsets move when other kernels grow)
        body.append(s)
    flush()
    return out


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "isa_baseline.json")
    fp = fingerprints()
    if mode == "save":
        os.makedirs(os.path.dirname(path), exist_ok=True)
        json.dump(fp, open(path, "w"), indent=0, sort_keys=True)
        print(f"{len(fp)} functions -> {path}")
        return 0
    base = json.load(open(path))
    changed = [k for k in base if k in fp and fp[k]["code"] != base[k]["code"]]
    gone = [k for k in base if k not in fp]
    new = [k for k in fp if k not in base]
    # a kernel whose template signature grew a defaulted parameter has a new mangled name and the same code: pair them by hash
    renamed = []
    for k in list(gone):
        m = [n for n in new if fp[n]["code"] == base[k]["code"] and n.split("I")[0] == k.split("I")[0]]
        if m:
            renamed.append((k, m[0]))
            gone.remove(k)
            new.remove(m[0])
    dem = subprocess.run(["c++filt"] + changed + gone + new, capture_output=True, text=True).stdout.splitlines() if (changed or gone or new) else []
    it = iter(dem)
    for tag, lst in (("CHANGED", changed), ("GONE", gone), ("new", new)):
        for k in lst:
            d = next(it)
            extra = f"  insns {base[k]['insns']} -> {fp[k]['insns']}" if tag == "CHANGED" else ""
            print(f"{tag:8s} {re.sub(r'[(].*', '', d)[:110]}{extra}")
    for k, n in renamed:
        d = subprocess.run(["c++filt", k, n], capture_output=True, text=True).stdout.splitlines()
        print(f"renamed  {re.sub(r'[(].*', '', d[0])[:70]} -> {re.sub(r'[(].*', '', d[1])[:70]} (same code)")
    print(f"{len(base) - len(changed) - len(gone)} of {len(base)} functions unchanged ({len(renamed)} of them under a new name), "
          f"{len(changed)} changed, {len(gone)} gone, {len(new)} new")
    return 1 if changed or gone else 0


if __name__ == "__main__":
    sys.exit(main())
