#!/bin/bash
# SQ counter summaries per kernel of the training step (the 37 / 36 / 28 % split of the fused kernels must be
# checkable from tracked files).  Two rocprofv3 --pmc passes (8 SQ slots each) with --kernel-trace only, eager steps of the bench
# configuration (tools/pmc_workload.py) → gpurun_out/sq_counters.txt; copy into profiles/r03_sq_counters.txt.
cd "$(dirname "$0")/.."; R=$PWD; mkdir -p gpurun_out
B=${1:-512}
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
i=0
for CNT in "$P1" "$P2"; do
  i=$((i+1)); rm -rf $R/gpurun_out/sq_p$i
  timeout 1200 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $R/gpurun_out/sq_p$i -- python $R/tools/pmc_workload.py --batch $B --steps 1 --mode train > $R/gpurun_out/sq_p$i.log 2>&1
  echo "pass $i rc=$?"; tail -2 $R/gpurun_out/sq_p$i.log
done
cd $R && python - <<'PY' | tee gpurun_out/sq_counters.txt
import collections, csv, glob, json, re, sys
sys.path.insert(0, ".")
import bench
bi = bench.build_info()
print("# SQ counters per kernel, summed over the dispatches of ONE eager training step at B = 512 (tools/pmc_workload.py)")
print("# source_hash", bi["source_hash"], "git", bi.get("git_sha"), "flags", json.dumps(bi["flags"]))
print("# WAVE_CYCLES / WAIT_* / ACTIVE_INST_* are in quad-cycles summed over waves; shares below are of SQ_WAVE_CYCLES")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob("gpurun_out/sq_p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        m = re.search(r"mlp_fused_kernel(?:_sf)?<\s*(\d+),\s*\d+,\s*\d+,\s*(\d)>", kn) or re.search(r"mlp_fused_kernel(?:_sf)?ILi(\d+)ELi\d+ELi\d+ELi(\d)EE", kn)
        if m: k = f"mlp_fused<C={m.group(1)},MODE={m.group(2)}>"
        elif "gemm_nt2" in kn: k = "gemm_nt2 (256 x BN, LDS-DMA)"
        elif "gemm_nt_fast" in kn: k = "gemm_nt_fast (128 x 128)"
        elif "gemm_tn_fast" in kn: k = "gemm_tn_fast"
        elif "dwconv7_mfma" in kn: k = "dwconv7_mfma (fwd / dgrad)"
        elif "dwconv7_wgrad_mfma" in kn: k = "dwconv7_wgrad_mfma"
        elif "ln_fwd" in kn or "ln_bwd" in kn: k = "layernorm fwd / bwd"
        else: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
for k in sorted(agg):
    d = agg[k]; wc = max(d.get("SQ_WAVE_CYCLES", 0), 1)
    sh = lambda n: f"{100 * d.get(n, 0) / wc:5.1f}%"
    print(f"{k:34s} dispatches {len(disp[k]):3d}  wait_any {sh('SQ_WAIT_ANY')}  wait_inst {sh('SQ_WAIT_INST_ANY')}  active_inst {sh('SQ_ACTIVE_INST_ANY')}"
          f"  valu {sh('SQ_ACTIVE_INST_VALU')}  lds {sh('SQ_ACTIVE_INST_LDS')}  | lds_bank_conflict / lds_idx_active "
          f"{100 * d.get('SQ_LDS_BANK_CONFLICT', 0) / max(d.get('SQ_LDS_IDX_ACTIVE', 0), 1):5.1f}%  insts valu/lds/mfma "
          f"{d.get('SQ_INSTS_VALU', 0):.3g}/{d.get('SQ_INSTS_LDS', 0):.3g}/{d.get('SQ_INSTS_MFMA', 0):.3g}  mfma_busy_cycles {d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):.3g}")
PY
rm -rf gpurun_out/sq_p1 gpurun_out/sq_p2
