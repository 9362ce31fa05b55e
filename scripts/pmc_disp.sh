#!/bin/bash
# usage: scripts/pmc_disp.sh "<counters>" <tag> <python script + args>  → per-DISPATCH counters (in launch order) for kernels matching $PMC_FILTER
cd "$(dirname "$0")/.."; R=$PWD; mkdir -p gpurun_out
CNT="$1"; TAG="$2"; shift 2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $R/gpurun_out/pmcd_$TAG -- python "$@" > $R/gpurun_out/pmcd_$TAG.log 2>&1
echo "rc=$?"
f=$(find $R/gpurun_out/pmcd_$TAG -name "*counter_collection.csv" | head -1)
python - "$f" "${PMC_FILTER:-gemm}" <<'PY'
import csv, sys, collections
f, filt = sys.argv[1], sys.argv[2]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if filt not in r["Kernel_Name"]:
        continue
    k = (int(r["Dispatch_Id"]), r["Kernel_Name"][:48], r["Grid_Size"])
    rows.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k, d in rows.items():
    print(k[0], k[1], "grid", k[2], {a: int(b) for a, b in d.items()})
PY
rm -rf $R/gpurun_out/pmcd_$TAG
