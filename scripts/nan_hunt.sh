#!/bin/bash
# Estimate how often a short training run of bench.py ends with a non-finite loss, per configuration.
# Usage (on a GPU box, through gpurun):  NRUN=14 bash scripts/nan_hunt.sh
# Round 1 observed (10-25 step runs): ~1 in 20-50 with VSX_FLAGS=nt_stream=3,grn_stream=2,ln_stream=3, none in the shipped
# defaults measured so far; see DESIGN.md §3 item 8 / §7.  Each run costs ~12 s.
cd "$(dirname "$0")/.."
NRUN=${NRUN:-14}
STEPS=${STEPS:-20}
run() {  # label, then VAR=value ... passed to env
  label=$1; shift
  n=0; bad=0
  for i in $(seq 1 "$NRUN"); do
    out=$(env "$@" python bench.py --no-cpu-baseline --steps "$STEPS" --warmup 2 2>/dev/null |
          python -c "import json,sys; print(json.loads(sys.stdin.read())['whole_path']['loss_finite'])")
    n=$((n + 1)); [ "$out" == "True" ] || bad=$((bad + 1))
  done
  echo "$label: $bad / $n runs with a non-finite loss"
}
run defaults            X=1
run nt_stream_stores    VSX_FLAGS=nt_stream=1
run nt_stream_aux_load  VSX_FLAGS=nt_stream=2
run grn_stream_load     VSX_FLAGS=grn_stream=2
run ln_stream_loads     VSX_FLAGS=ln_stream=3
