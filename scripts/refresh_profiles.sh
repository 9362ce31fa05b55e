#!/bin/bash
# One GPU-box visit that regenerates everything under profiles/ for the current code state (round prefix R, default r04):
#   rocprofv3 kernel trace + stats of the bench AND of N plain training steps (tools/pmc_workload.py: the per-family table of
#   tools/roofline_table.py needs a trace that holds training steps only), PMC traffic passes (FETCH_SIZE / WRITE_SIZE), L2
#   hit / miss counters of the GEMM kernels, the un-profiled default bench line.
# Run through gpurun, then `bash scripts/refresh_profiles.sh --collect` locally copies gpurun_out/* into profiles/.
cd "$(dirname "$0")/.."
R=${ROUND:-r06}
if [ "$1" == "--collect" ]; then
  cp gpurun_out/kernel_stats.csv profiles/${R}_kernel_stats.csv
  cp gpurun_out/kernel_stats_train.csv profiles/${R}_kernel_stats_train.csv
  cp gpurun_out/by_kernel_and_grid.txt profiles/${R}_by_kernel_and_grid.txt
  cp gpurun_out/prof_bench.json profiles/${R}_bench_under_rocprof.json
  cp gpurun_out/pmc_traffic.json profiles/${R}_pmc_traffic_b512.json
  cp gpurun_out/pmc_traffic.txt profiles/${R}_pmc_traffic_b512.txt
  cp gpurun_out/pmc_traffic_fwd.json profiles/${R}_pmc_traffic_fwd_b512.json
  cp gpurun_out/pmc_traffic_fwd.txt profiles/${R}_pmc_traffic_fwd_b512.txt
  cp gpurun_out/bench_default.json profiles/${R}_bench_default.json
  cp gpurun_out/tcc_gemm.txt profiles/${R}_tcc_gemm.txt
  cp gpurun_out/roofline_table.txt profiles/${R}_roofline_table.txt
  exit 0
fi
STEPS_TRAIN=3
bash scripts/gpu_check.sh prof > /dev/null 2>&1
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/kernel_stats.csv
t=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1); python tools/ktrace.py "$t" > gpurun_out/by_kernel_and_grid.txt 2>&1
rm -rf gpurun_out/prof
# kernel stats of STEPS_TRAIN eager training steps and nothing else
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_train" -- \
    python "$GRAFT_REPO_ROOT/tools/pmc_workload.py" --batch 512 --steps $STEPS_TRAIN --mode train > "$GRAFT_REPO_ROOT/gpurun_out/prof_train.log" 2>&1 )
f=$(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/kernel_stats_train.csv
rm -rf gpurun_out/prof_train
bash scripts/pmc_traffic.sh 512 2 train pmc_traffic > /dev/null 2>&1
bash scripts/pmc_traffic.sh 512 3 fwd pmc_traffic_fwd > /dev/null 2>&1
bash scripts/tcc_gemm.sh > /dev/null 2>&1
# the bench line needs the traffic file of THIS source state in place to attach roofline.traffic
cp gpurun_out/pmc_traffic.json profiles/${R}_pmc_traffic_b512.json
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python tools/roofline_table.py gpurun_out/kernel_stats_train.csv $STEPS_TRAIN gpurun_out/pmc_traffic.json gpurun_out/bench_default.json > gpurun_out/roofline_table.txt 2>&1
cat gpurun_out/bench_default.json
