"""Wave-quantisation view of a rocprofv3 kernel_trace.csv: for every (kernel, grid) the workgroups per launch, the workgroups a CU
can hold (from the trace's VGPR / accumulator-VGPR / LDS / workgroup-size columns: 512 registers per SIMD lane in 8-register
granules, 160 KiB of LDS, 32 waves per CU), the number of "rounds" the launch needs on 256 CUs and the fraction of the last
round that is filled.  A launch of 1.12 rounds runs its last eighth at a fraction of the chip.

    python tools/rounds.py <kernel_trace.csv> [steps] [min_ms_per_step]
"""
import collections
import csv
import math
import re
import sys

f = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
min_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.2
CUS = 256
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    name = re.sub(r"^void ", "", r["Kernel_Name"])
    m = re.match(r"_Z\d+([A-Za-z0-9_]+?)I", name)
    short = m.group(1) if m else name.split("(")[0][:60]
    if name.startswith("_Z"):
        short += "<" + "".join(re.findall(r"(Li\d+E|Lb\dE)", name.split("Ev")[0])).replace("Li", "").replace("Lb", "b").replace("E", ",") + ">"
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    vg = 2 * int(r.get("VGPR_Count", 0) or 0) + int(r.get("Accum_VGPR_Count", 0) or 0)  # (the trace counts arch VGPRs in pairs: 128 = a 256-register kernel)
    lds = int(r.get("LDS_Block_Size", 0) or 0)
    key = (short, grid // wg, wg, vg, lds)
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += d
rows = []
for (k, nwg, wg, vg, lds), (n, t) in agg.items():
    waves = (wg + 63) // 64
    alloc = max(8, (vg + 7) // 8 * 8)
    wps = min(8, 512 // alloc)                       # waves per SIMD the registers allow
    by_reg = (wps * 4) // waves if waves <= wps * 4 else 0
    by_lds = (160 * 1024) // lds if lds else 99
    by_waves = 32 // waves
    per_cu = max(1, min(by_reg, by_lds, by_waves))
    rounds = nwg / (CUS * per_cu)
    eff = rounds / math.ceil(rounds) if rounds > 0 else 1.0
    rows.append((t / steps / 1e3, k, nwg, wg, vg, lds, per_cu, rounds, eff, n / steps, t / n))
rows.sort(reverse=True)
print(f"{'kernel':58s} {'WGs':>7s} {'thr':>5s} {'regs':>4s} {'LDS':>7s} {'WG/CU':>5s} {'rounds':>7s} {'fill':>5s} {'n/step':>6s} {'avg us':>8s} {'ms/step':>7s}")
for ms, k, nwg, wg, vg, lds, per_cu, rounds, eff, n, avg in rows:
    if ms < min_ms:
        continue
    print(f"{k[:58]:58s} {nwg:7d} {wg:5d} {vg:4d} {lds:7d} {per_cu:5d} {rounds:7.2f} {eff:5.2f} {n:6.1f} {avg:8.1f} {ms:7.2f}")
