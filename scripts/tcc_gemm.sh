#!/bin/bash
# Where does the operand stream of the GEMM kernels come from?    L2 hit / miss and L1 -> L2 read requests per
# GEMM kernel over ONE eager training step at B = 512 (tools/pmc_workload.py), in their own rocprofv3 --pmc pass (kernel trace only).
#   -> gpurun_out/tcc_gemm.txt ; copy into profiles/r04_tcc_gemm.txt
cd "$(dirname "$0")/.."; R=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tcc_p
timeout 1200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $R/gpurun_out/tcc_p -- \
  python $R/tools/pmc_workload.py --batch 512 --steps 1 --mode train > $R/gpurun_out/tcc_p.log 2>&1
echo "tcc pass rc=$?"; tail -2 $R/gpurun_out/tcc_p.log
cd $R && python - <<'PY' | tee gpurun_out/tcc_gemm.txt
import collections, csv, glob, json, re, sys
sys.path.insert(0, ".")
import bench
bi = bench.build_info()
print("# L2 (TCC) hit / miss requests and L1 -> L2 read requests per kernel, summed over the dispatches of ONE eager training step at B = 512")
print("# source_hash", bi["source_hash"], "git", bi.get("git_sha"))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob("gpurun_out/tcc_p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        fam = bench.kernel_family(kn)
        if not fam.startswith("gemm") and fam not in ("mlp_fused", "dwconv7"):
            continue
        m = re.search(r"(gemm_nt2_lnbwd_kernel|gemm_nt2_kernel|gemm_nt_fast_kernel|gemm_nt_kernel|gemm_tn_fast_kernel|gemm_tn_kernel|mlp_fused_kernel|dwconv7_mfma_kernel|dwconv7_wgrad_mfma_kernel)", kn)
        k = m.group(1) if m else fam
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
for k in sorted(agg):
    d = agg[k]; h, ms = d.get("TCC_HIT_sum", 0.0), d.get("TCC_MISS_sum", 0.0)
    print(f"{k:28s} dispatches {len(disp[k]):3d}  TCC_HIT {h:.4g}  TCC_MISS {ms:.4g}  hit rate {100 * h / max(h + ms, 1):5.1f}%  TCP_TCC_READ_REQ {d.get('TCP_TCC_READ_REQ_sum', 0):.4g}")
PY
rm -rf gpurun_out/tcc_p
