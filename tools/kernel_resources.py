"""Per-kernel register / spill / LDS summary of csrc/psgdk.hip (hipcc -Rpass-analysis=kernel-resource-usage), filtered by a
substring: `python tools/kernel_resources.py lra_`."""
import re
import subprocess
import sys

pat = sys.argv[1] if len(sys.argv) > 1 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-Rpass-analysis=kernel-resource-usage",
                      "psgd_torch_amd/csrc/psgdk.hip", "-o", "/tmp/_kr.o"], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[^\]]*\])?): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
demangle = subprocess.run(["c++filt"] + list(rows), capture_output=True, text=True).stdout.splitlines()
print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'LDS':>7s} {'occ':>4s}")
for k, d in zip(demangle, rows.values()):
    if pat in k:
        name = re.sub(r"\(.*", "", k)[:70]
        print(f"{name:70s} {d.get('VGPRs', 0):5d} {d.get('AGPRs', 0):5d} {d.get('SGPRs', 0):5d} {d.get('VGPRs Spill', 0):6d} "
              f"{d.get('SGPRs Spill', 0):6d} {d.get('ScratchSize [bytes/lane]', 0):7d} {d.get('LDS Size [bytes/block]', 0):7d} "
              f"{d.get('Occupancy [waves/SIMD]', 0):4d}")
