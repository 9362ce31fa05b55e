#!/bin/bash
# One GPU-box visit that regenerates everything under profiles/ for the current code state:
#   rocprofv3 kernel trace + stats of the bench, PMC traffic passes, the un-profiled default bench line.
# Run through gpurun, then `bash scripts/refresh_profiles.sh --collect` locally copies gpurun_out/* into profiles/.
cd "$(dirname "$0")/.."
if [ "$1" == "--collect" ]; then
  cp gpurun_out/kernel_stats.csv profiles/r01_final_kernel_stats.csv
  cp gpurun_out/by_kernel_and_grid.txt profiles/r01_final_by_kernel_and_grid.txt
  cp gpurun_out/prof_bench.json profiles/r01_final_bench_under_rocprof.json
  cp gpurun_out/pmc_traffic.json profiles/r01_pmc_traffic_b512.json
  cp gpurun_out/pmc_traffic.txt profiles/r01_pmc_traffic_b512.txt
  cp gpurun_out/bench_default.json profiles/r01_final_bench_default.json
  exit 0
fi
bash scripts/gpu_check.sh prof > /dev/null 2>&1
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/kernel_stats.csv
t=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1); python tools/ktrace.py "$t" > gpurun_out/by_kernel_and_grid.txt 2>&1
rm -rf gpurun_out/prof
bash scripts/pmc_traffic.sh 512 2 > /dev/null 2>&1
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cat gpurun_out/bench_default.json
