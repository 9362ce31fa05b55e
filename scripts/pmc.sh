#!/bin/bash
# usage: scripts/pmc.sh "<counters>" <tag> <python args...>   → gpurun_out/pmc_<tag>.csv (per-dispatch counters)
cd "$(dirname "$0")/.."; R=$PWD; mkdir -p gpurun_out
CNT="$1"; TAG="$2"; shift 2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG -- python "$@" > $R/gpurun_out/pmc_$TAG.log 2>&1
echo "rc=$?"
f=$(find $R/gpurun_out/pmc_$TAG -name "*counter_collection.csv" | head -1)
echo "file: $f"
python - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == list(agg[k].keys())[0]: cnt[k] += 1
for k, d in agg.items():
    print(k, "dispatches", cnt[k], {a: round(b / max(cnt[k], 1)) for a, b in d.items()})
PY
