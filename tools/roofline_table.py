"""Per-kernel-family roofline table of the training step, from TRACKED files only:

    python tools/roofline_table.py profiles/r04_kernel_stats_train.csv 3 profiles/r04_pmc_traffic_b512.json [profiles/r04_bench_default.json]

* kernel_stats CSV: `rocprofv3 --kernel-trace --stats` of `tools/pmc_workload.py --batch 512 --steps N --mode train` (N eager
  training steps of the bench configuration and nothing else; the two calibration launches are dropped) -> ms per step;
* PMC traffic JSON (scripts/pmc_traffic.sh): FETCH_SIZE / WRITE_SIZE passes of the same workload -> HBM-side GB per step;
* bench JSON (optional): the live `roofline_classes` of bench.py (HIP events on the launch stream) -> algorithmic bytes / flops,
  printed beside the trace-side numbers so that the two clocks can be compared.
Families = bench.KERNEL_FAMILY (one per kernel template / kernel group)."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import HBM_PEAK_GBS, MFMA_BF16_PEAK_TFLOPS, kernel_family  # noqa: E402


def main(stats_csv, steps, pmc_json, bench_json=None):
    fam = collections.defaultdict(lambda: {"ms": 0.0, "launches": 0})
    for r in csv.DictReader(open(stats_csv)):
        if "normalize_kernel" in r["Name"]:
            continue
        f = kernel_family(r["Name"])
        fam[f]["ms"] += float(r["TotalDurationNs"]) / 1e6 / steps
        fam[f]["launches"] += int(r["Calls"]) / steps
    pmc = json.load(open(pmc_json))
    tf = pmc.get("families", {})
    live = {}
    if bench_json:
        bj = json.load(open(bench_json))
        live = {c["family"]: c for c in bj.get("roofline_classes", [])}
    tot = sum(v["ms"] for v in fam.values())
    print(f"# kernel time {tot:.2f} ms / step over {steps} eager steps; PMC file: source_hash {pmc.get('source_hash')}, "
          f"{pmc.get('total_GB_per_step', 0):.1f} GB / step = {pmc.get('total_MB_per_patch', 0):.1f} MB / patch")
    print(f"{'family':18s} {'ms/step':>8s} {'%':>5s} {'launch':>6s} {'R GB':>7s} {'W GB':>7s} {'traffic TB/s':>12s} {'frac 8TB/s':>10s} | "
          f"{'live ms':>8s} {'algo GB':>8s} {'hbm_frac':>8s} {'mfma_frac':>9s}")
    for f, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"]):
        t = tf.get(f, {})
        gb = t.get("read_GB_per_step", 0.0) + t.get("write_GB_per_step", 0.0)
        tbs = gb / max(v["ms"], 1e-9)
        lv = live.get(f)
        extra = (f"{lv['ms_per_step']:8.2f} {lv['algorithmic_GB_per_step']:8.2f} {lv['hbm_frac']:8.3f} {lv['mfma_frac']:9.3f}" if lv else "")
        print(f"{f:18s} {v['ms']:8.2f} {100 * v['ms'] / tot:5.1f} {v['launches']:6.0f} {t.get('read_GB_per_step', 0.0):7.2f} "
              f"{t.get('write_GB_per_step', 0.0):7.2f} {tbs:12.2f} {tbs * 1e3 / HBM_PEAK_GBS:10.3f} | {extra}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None)
