#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r2_gpu_tests5.log
timeout 900 python bench.py > gpurun_out/r2_bench5.json 2> gpurun_out/r2_bench5.err
