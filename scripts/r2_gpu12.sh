#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "fused" 2>&1 | tail -6 > gpurun_out/r2_tests12.log
timeout 600 python tools/perf_mlp.py 512 > gpurun_out/r2_perf_mlp12.log 2>&1
VSX_FLAGS=mlp_fused=15 timeout 900 python bench.py --no-cpu-baseline --no-gate > gpurun_out/r2_bench12_f15.json 2> gpurun_out/r2_bench12_f15.err
VSX_FLAGS=mlp_fused=11 timeout 900 python bench.py --no-cpu-baseline --no-gate > gpurun_out/r2_bench12_f11.json 2> gpurun_out/r2_bench12_f11.err
