timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "ln or layernorm or norm" 2>&1 | tail -3
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/r3_b9.json 2> gpurun_out/r3_b9.err; cut -c1-260 gpurun_out/r3_b9.json; grep -E "^\[ops\] (ln_|dgrad_ln)" gpurun_out/r3_b9.err
