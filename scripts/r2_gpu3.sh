#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "fused_grn_mlp" 2>&1 | tail -15 > gpurun_out/r2_mlp_test.log
timeout 600 python tools/perf_mlp.py 512 > gpurun_out/r2_perf_mlp.log 2>&1
timeout 1800 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_soak.py 2>&1 | tail -15 > gpurun_out/r2_gpu_tests3.log
timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r2_bench3.json 2> gpurun_out/r2_bench3.err
