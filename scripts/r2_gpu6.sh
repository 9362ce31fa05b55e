#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -m gpu -k "stem or golden or oracle or bf16_tracks or training_steps" 2>&1 | tail -5 > gpurun_out/r2_tests6.log
timeout 600 python bench.py --no-cpu-baseline --no-gate --steps 10 --warmup 3 > gpurun_out/r2_bench6.json 2> gpurun_out/r2_bench6.err
timeout 600 python bench.py --force-dp --no-cpu-baseline --no-gate --steps 10 --warmup 3 > gpurun_out/r2_bench6_dp.json 2> gpurun_out/r2_bench6_dp.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-gate > gpurun_out/r2_bench6_tr.json 2> gpurun_out/r2_bench6_tr.err
