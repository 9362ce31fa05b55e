#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "fused" 2>&1 | tail -8 > gpurun_out/r2_mlp_test8.log
timeout 600 python tools/perf_mlp.py 512 > gpurun_out/r2_perf_mlp8.log 2>&1
