#!/bin/bash
# One GPU-box visit: op-level parity, model-level parity, smoke, bench (+ per-op table).  Logs → gpurun_out/
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
STAGES="${1:-ops model smoke bench}"
for s in $STAGES; do
  case $s in
    ops)   timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --maxfail=40 -x --timeout=600 > gpurun_out/ops.log 2>&1; echo "ops rc=$?" ; tail -25 gpurun_out/ops.log ;;
    opsall) timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -q --maxfail=60 --timeout=600 > gpurun_out/ops.log 2>&1; echo "ops rc=$?" ; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/ops.log | tail -70 ;;
    model) timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -q --maxfail=20 -s --timeout=900 > gpurun_out/model.log 2>&1; echo "model rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|worst|Error|assert" gpurun_out/model.log | tail -40 ;;
    boundary) timeout 900 python -m pytest tests/test_gpu_boundary.py -m gpu -q --maxfail=20 --timeout=600 > gpurun_out/boundary.log 2>&1; echo "boundary rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|Error|assert" gpurun_out/boundary.log | tail -40 ;;
    smoke) timeout 600 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -5 gpurun_out/smoke.log ;;
    bench) timeout 900 python bench.py --steps 10 --warmup 3 --profile-ops > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cat gpurun_out/bench.json; tail -30 gpurun_out/bench.err ;;
    prof)  cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof.err"; echo "prof rc=$?"; cd "$GRAFT_REPO_ROOT"; find gpurun_out/prof -name "*kernel_stats*" | head; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" ;;
  esac
done
