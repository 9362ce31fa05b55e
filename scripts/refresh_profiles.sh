#!/bin/bash
# One GPU-box visit that regenerates everything under profiles/ for the current code state:
#   rocprofv3 kernel trace + stats of the bench, PMC traffic passes, the un-profiled default bench line.
# Run through gpurun, then `bash scripts/refresh_profiles.sh --collect` locally copies gpurun_out/* into profiles/.
cd "$(dirname "$0")/.."
if [ "$1" == "--collect" ]; then
  R=${ROUND:-r03}
  cp gpurun_out/kernel_stats.csv profiles/${R}_kernel_stats.csv
  cp gpurun_out/by_kernel_and_grid.txt profiles/${R}_by_kernel_and_grid.txt
  cp gpurun_out/prof_bench.json profiles/${R}_bench_under_rocprof.json
  cp gpurun_out/pmc_traffic.json profiles/${R}_pmc_traffic_b512.json
  cp gpurun_out/pmc_traffic.txt profiles/${R}_pmc_traffic_b512.txt
  cp gpurun_out/pmc_traffic_fwd.json profiles/${R}_pmc_traffic_fwd_b512.json
  cp gpurun_out/pmc_traffic_fwd.txt profiles/${R}_pmc_traffic_fwd_b512.txt
  cp gpurun_out/bench_default.json profiles/${R}_bench_default.json
  exit 0
fi
bash scripts/gpu_check.sh prof > /dev/null 2>&1
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/kernel_stats.csv
t=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1); python tools/ktrace.py "$t" > gpurun_out/by_kernel_and_grid.txt 2>&1
rm -rf gpurun_out/prof
bash scripts/pmc_traffic.sh 512 2 train pmc_traffic > /dev/null 2>&1
bash scripts/pmc_traffic.sh 512 3 fwd pmc_traffic_fwd > /dev/null 2>&1
# the bench line needs the traffic file of THIS source state in place to attach roofline.traffic
cp gpurun_out/pmc_traffic.json profiles/r03_pmc_traffic_b512.json
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cat gpurun_out/bench_default.json
