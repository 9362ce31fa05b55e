for i in 1 2 3 4; do timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "baseline_size" 2>&1 | grep -E "^E  |passed|failed" | cut -c1-300; done
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --profile-ops > gpurun_out/r3_b11.json 2> gpurun_out/r3_b11.err; cut -c1-260 gpurun_out/r3_b11.json
