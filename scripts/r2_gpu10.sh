#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r2_gpu_tests10.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r2_bench10.json 2> gpurun_out/r2_bench10.err
VSX_FLAGS=mlp_fused=3 timeout 900 python bench.py --no-cpu-baseline --no-gate > gpurun_out/r2_bench10_f3.json 2> gpurun_out/r2_bench10_f3.err
