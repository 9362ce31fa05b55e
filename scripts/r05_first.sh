#!/bin/bash
# round-5 baseline visit: whole -m gpu suite, default bench line, kernel stats of 3 plain training steps
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x --timeout=900 > gpurun_out/gpu_all.log 2>&1; echo "gpu tests rc=$?"; tail -5 gpurun_out/gpu_all.log
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"; cat gpurun_out/bench_default.json
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_train" -- \
    python "$GRAFT_REPO_ROOT/tools/pmc_workload.py" --batch 512 --steps 3 --mode train > "$GRAFT_REPO_ROOT/gpurun_out/prof_train.log" 2>&1 )
f=$(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/kernel_stats_train.csv
t=$(find gpurun_out/prof_train -name "*kernel_trace.csv" | head -1); python tools/ktrace.py "$t" > gpurun_out/by_kernel_and_grid_train.txt 2>&1
rm -rf gpurun_out/prof_train
head -50 gpurun_out/kernel_stats_train.csv
