#!/bin/bash
# round-2, second GPU pass: the refactored step / optimiser under the whole GPU suite, then a longer bench-style NaN hunt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r2_gpu_tests.log
hunt() { label=$1; shift; n=0; bad=0
  for i in $(seq 1 "$NRUN"); do
    out=$(env "$@" timeout 120 python bench.py --no-cpu-baseline --steps 20 --warmup 2 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['whole_path']['loss_finite'])" 2>/dev/null)
    n=$((n + 1)); [ "$out" == "True" ] || bad=$((bad + 1)); done
  echo "$label: $bad / $n runs with a non-finite loss"; }
NRUN=30 hunt all_stream_flags VSX_FLAGS=nt_stream=3,grn_stream=2,ln_stream=3 > gpurun_out/r2_nan_hunt2.log 2>&1
NRUN=15 hunt defaults X=1 >> gpurun_out/r2_nan_hunt2.log 2>&1
