"""Summarise a rocprofv3 kernel_trace.csv: per (kernel, grid) count / avg / total us (compact names)."""
import csv, sys, re, collections
f = sys.argv[1]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_Z\d+([A-Za-z0-9_]+?)I", name)
    short = m.group(1) if m else name.split("(")[0][:48]
    if name.startswith("_Z"):
        tparams = "".join(re.findall(r"(Li\d+E|Lb\dE)", name.split("Ev")[0]))
        short += "<" + ("bf16" if "DF16b" in name else "f32") + " " + tparams + ">"
    key = (short, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1; a[1] += d
tot = sum(v[1] for v in agg.values())
order = sorted(agg.items(), key=lambda kv: -kv[1][1]) if "--sort" in sys.argv else agg.items()
for (k, gx, gy, gz), (n, t) in order:
    print(f"{k[:60]:60s} grid=({gx},{gy},{gz}) n={n:4d} avg={t/n:9.1f} us total={t/1e3:8.2f} ms {100*t/tot:5.1f}%")
print(f"TOTAL kernel time {tot/1e3:.2f} ms")
