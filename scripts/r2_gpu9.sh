#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
VSX_FLAGS=mlp_fused=11 timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r2_gpu_tests9.log
VSX_FLAGS=mlp_fused=11 timeout 900 python bench.py --no-cpu-baseline --no-gate > gpurun_out/r2_bench9_f11.json 2> gpurun_out/r2_bench9_f11.err
VSX_FLAGS=mlp_fused=3 timeout 900 python bench.py --no-cpu-baseline --no-gate > gpurun_out/r2_bench9_f3.json 2> gpurun_out/r2_bench9_f3.err
VSX_FLAGS=mlp_fused=11 timeout 900 python bench.py --no-cpu-baseline --no-gate > gpurun_out/r2_bench9_f11b.json 2> gpurun_out/r2_bench9_f11b.err
