"""Turn the two rocprofv3 counter_collection.csv files (FETCH_SIZE pass, WRITE_SIZE pass) of tools/pmc_workload.py
into per-kernel HBM traffic per launch, calibrated on the known-byte-count launches (MI355X_MICROARCH.md, HBM section:
gfx950's FETCH_SIZE needs a correction, WRITE_SIZE is uncalibrated → calibrate in the own access pattern)."""
import collections
import csv
import json
import re
import sys


def load(path, counter):
    rows = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            kn = r["Kernel_Name"]
            m = re.search(r"mlp_fused_kernel(?:_sf)?<\s*\d+,\s*\d+,\s*\d+,\s*(\d)>", kn) or re.search(r"mlp_fused_kernel(?:_sf)?ILi\d+ELi\d+ELi\d+ELi(\d)EE", kn)
            if m:  # the pass (template MODE) is what the op classes of bench.py distinguish
                rows.append((f"mlp_fused_kernel_mode{m.group(1)}", float(r["Counter_Value"])))
                continue
            kn = kn.replace("(anonymous namespace)::", "")  # gemm_nt2_kernel lives in an anonymous namespace
            rows.append((re.sub(r"[<(].*", "", kn).replace("void ", "").strip(), float(r["Counter_Value"])))
    return rows


def load_templates(path, counter):
    """(template, counts) rows keyed the way bench.py names a kernel TEMPLATE in `roofline.kernel` — the fused GRN-MLP passes by
    width and MODE (their names carry both; a GEMM's shape is not in its kernel name, so GEMM templates have no row here)"""
    rows = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        kn = r["Kernel_Name"]
        m = re.search(r"mlp_fused_kernel(?:_sf)?<\s*(\d+),\s*\d+,\s*\d+,\s*(\d)>", kn) or re.search(r"mlp_fused_kernel(?:_sf)?ILi(\d+)ELi\d+ELi\d+ELi(\d)EE", kn)
        if m:
            rows.append((f"mlp_fused_kernel<C={m.group(1)}, MODE={m.group(2)}>", float(r["Counter_Value"])))
    return rows


def main(fetch_csv, write_csv, steps, batch, out):
    f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    cal = {}
    for name, rows, key in (("FETCH_SIZE", f, "read"), ("WRITE_SIZE", w, "write")):
        vals = [v for k, v in rows if k == "normalize_kernel"]
        # launches: 2 x 256 MiB, 2 x 1024 MiB
        assert len(vals) == 4, (name, len(vals))
        per_mib = [vals[0] / 256, vals[1] / 256, vals[2] / 1024, vals[3] / 1024]
        cal[key] = {"counter_per_MiB": per_mib, "bytes_per_count": (1 << 20) / (sum(per_mib[2:]) / 2)}
    agg = collections.defaultdict(lambda: {"launches": 0, "read": 0.0, "write": 0.0})
    for k, v in f:
        agg[k]["launches"] += 1
        agg[k]["read"] += v * cal["read"]["bytes_per_count"]
    for k, v in w:
        agg[k]["write"] += v * cal["write"]["bytes_per_count"]
    res = {"steps": steps, "batch": batch, "calibration": cal, "kernels": {}}
    try:  # identity of what was measured: bench.py refuses a traffic file from other kernel sources / flags
        import os

        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench

        bi = bench.build_info()
        res["source_hash"], res["flags"], res["git_sha"] = bi["source_hash"], bi["flags"], bi.get("git_sha")
    except Exception as e:  # noqa: BLE001
        res["identity_error"] = repr(e)
    for k, d in sorted(agg.items(), key=lambda kv: -(kv[1]["read"] + kv[1]["write"])):
        if k == "normalize_kernel" or "distribution_elementwise" in k:
            continue  # the calibration launches and the torch.rand that fills their 256 MiB / 1 GiB inputs: not part of a step
        n = d["launches"]
        res["kernels"][k] = {"launches_per_step": n / steps, "read_GB_per_step": d["read"] / steps / 1e9, "write_GB_per_step": d["write"] / steps / 1e9,
                             "traffic_bytes_per_launch": (d["read"] + d["write"]) / n}
    # op classes as bench.py's OpTimer names them (all instantiations of one kernel family together)
    cls = collections.defaultdict(lambda: {"launches_per_step": 0.0, "read_GB_per_step": 0.0, "write_GB_per_step": 0.0})
    for k, v in res["kernels"].items():
        base = re.sub(r"^_Z\d+", "", k)
        base = re.sub(r"_kernel.*", "", base)
        base = {"gemm_nt_fast": "gemm_nt", "gemm_tn_fast": "gemm_tn"}.get(base, base)
        if "gemm_nt2" in base:  # second-generation NT kernel (anonymous namespace: _ZN12_GLOBAL__N_1...)
            base = "gemm_nt"
        if base.startswith("mlp_fused"):  # bench.py's OpTimer classes = the ops wrappers of the five passes
            base = {"0": "mlp_stats", "1": "mlp_out", "2": "mlp_fc1", "3": "mlp_bwd_stats", "4": "mlp_bwd_dh"}.get(k[-1], "mlp_fused")
        for f in ("launches_per_step", "read_GB_per_step", "write_GB_per_step"):
            cls[base][f] += v[f]
    for c in cls.values():
        c["traffic_bytes_per_launch"] = (c["read_GB_per_step"] + c["write_GB_per_step"]) * 1e9 / max(c["launches_per_step"], 1e-9)
    res["classes"] = dict(cls)
    # kernel FAMILIES as bench.py's roofline_classes names them (bench.KERNEL_FAMILY: one per kernel template / kernel group)
    fam = collections.defaultdict(lambda: {"launches_per_step": 0.0, "read_GB_per_step": 0.0, "write_GB_per_step": 0.0})
    try:
        from bench import kernel_family
    except Exception:  # noqa: BLE001
        kernel_family = None
    if kernel_family is not None:
        for k, v in res["kernels"].items():
            for f in ("launches_per_step", "read_GB_per_step", "write_GB_per_step"):
                fam[kernel_family(k)][f] += v[f]
    res["families"] = dict(fam)
    tmpl = collections.defaultdict(lambda: {"launches_per_step": 0.0, "read_GB_per_step": 0.0, "write_GB_per_step": 0.0})
    for k, v in load_templates(fetch_csv, "FETCH_SIZE"):
        tmpl[k]["launches_per_step"] += 1.0 / steps
        tmpl[k]["read_GB_per_step"] += v * cal["read"]["bytes_per_count"] / steps / 1e9
    for k, v in load_templates(write_csv, "WRITE_SIZE"):
        tmpl[k]["write_GB_per_step"] += v * cal["write"]["bytes_per_count"] / steps / 1e9
    res["templates"] = dict(tmpl)
    tot = sum(v["read_GB_per_step"] + v["write_GB_per_step"] for v in res["kernels"].values())
    res["total_GB_per_step"] = tot
    res["total_MB_per_patch"] = tot * 1e3 / batch
    json.dump(res, open(out, "w"), indent=1)
    print("calibration", json.dumps(cal))
    print(f"total {tot:.2f} GB/step = {tot * 1e3 / batch:.1f} MB/patch")
    for k, v in sorted(res["classes"].items(), key=lambda kv: -(kv[1]["read_GB_per_step"] + kv[1]["write_GB_per_step"])):
        print(f"[class] {k:28s} {v['launches_per_step']:6.1f}/step  R {v['read_GB_per_step']:7.3f} GB  W {v['write_GB_per_step']:7.3f} GB  {v['traffic_bytes_per_launch'] / 1e6:9.2f} MB/launch")
    for k, v in sorted(res["families"].items(), key=lambda kv: -(kv[1]["read_GB_per_step"] + kv[1]["write_GB_per_step"])):
        print(f"[family] {k:27s} {v['launches_per_step']:6.1f}/step  R {v['read_GB_per_step']:7.3f} GB  W {v['write_GB_per_step']:7.3f} GB")
    for k, v in list(res["kernels"].items())[:25]:
        print(f"{k:40s} {v['launches_per_step']:6.1f}/step  R {v['read_GB_per_step']:7.3f} GB  W {v['write_GB_per_step']:7.3f} GB  {v['traffic_bytes_per_launch'] / 1e6:9.2f} MB/launch")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
