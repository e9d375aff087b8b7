"""Registers / spills / LDS of every kernel of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_regs.py gdmcf_amd/csrc/gemm_f32.hip [substring]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", src, "-c", "-o",
                    "/tmp/_kr.o", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
    if "error" in line:
        print(line)
dem = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
for name, d in zip(dem, rows.values()):
    if flt in name:
        print(f"{d.get('VGPRs', -1):4d} vgpr {d.get('AGPRs', 0):4d} agpr {d.get('VGPRs Spill', 0):4d} spill  occ {d.get('Occupancy', -1)}  "
              f"lds {d.get('LDS Size', 0)}  {name[:150]}")
