"""Per-kernel resource table (registers, spill, LDS, occupancy) of one .hip file: hipcc -Rpass-analysis=kernel-resource-usage
    python tools/kres.py viscy_amd/csrc/mlp.hip [name filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/kres.o"]
p = subprocess.run(cmd, capture_output=True, text=True)
if p.returncode:
    print(p.stderr[-4000:])
    sys.exit(1)
cur = None
rows = {}
for line in p.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?:\s+(\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
dem = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
for name, d in zip(dem, rows.values()):
    if flt and flt not in name:
        continue
    print(f"{name[:70]:70s} VGPR {d.get('VGPRs', 0):4d} AGPR {d.get('AGPRs', 0):4d} spill {d.get('VGPRs Spill', 0):4d} scratch {d.get('ScratchSize', 0):5d} "
          f"LDS {d.get('LDS Size', 0):7d} occ {d.get('Occupancy', 0)}")
