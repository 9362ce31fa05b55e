#!/bin/bash
# round-2 NaN hunt, first GPU pass: poison / determinism / soak tests, then bench-style short runs (host racing ahead)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_soak.py -x -q -m gpu 2>&1 | tail -40 > gpurun_out/r2_soak_tests.log
for fl in "" "nt_stream=3,grn_stream=2,ln_stream=3"; do
  timeout 300 python tools/nan_soak.py --steps 400 --batch 128 --flags "$fl" 2>gpurun_out/r2_nansoak.err | tail -1 >> gpurun_out/r2_nansoak.jsonl
done
timeout 300 python tools/nan_soak.py --steps 60 --batch 512 2>>gpurun_out/r2_nansoak.err | tail -1 >> gpurun_out/r2_nansoak.jsonl
NRUN=8 STEPS=20 timeout 900 bash scripts/nan_hunt.sh > gpurun_out/r2_nan_hunt.log 2>&1
