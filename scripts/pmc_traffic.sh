#!/bin/bash
# HBM traffic per kernel from PMC counters: two separate passes (FETCH_SIZE, WRITE_SIZE do not fit one pass) with
# --kernel-trace only.  → gpurun_out/pmc_traffic.json (+ .txt); copy into profiles/ to have it judged / read by bench.py
cd "$(dirname "$0")/.."; R=$PWD; mkdir -p gpurun_out
B=${1:-128}; STEPS=${2:-2}; MODE=${3:-train}; OUT=${4:-pmc_traffic}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$C
  timeout 1200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$C -- python $R/tools/pmc_workload.py --batch $B --steps $STEPS --mode $MODE > $R/gpurun_out/pmc_$C.log 2>&1
  echo "$C rc=$?"; tail -2 $R/gpurun_out/pmc_$C.log
done
F=$(find $R/gpurun_out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find $R/gpurun_out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
cd $R && python tools/pmc_traffic.py "$F" "$W" $STEPS $B gpurun_out/$OUT.json | tee gpurun_out/$OUT.txt
# the raw per-dispatch CSVs are large: keep only the summaries
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
