#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -x -q -m gpu -k "fused or gate_shape" 2>&1 | tail -6 > gpurun_out/r2_tests11.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r2_bench11.json 2> gpurun_out/r2_bench11.err
