bash scripts/refresh_profiles.sh
bash scripts/sq_counters.sh > gpurun_out/sq_run.log 2>&1
timeout 900 python bench.py --force-dp --no-cpu-baseline > gpurun_out/bench_force_dp.json 2> gpurun_out/bench_force_dp.err
STEPS=10 timeout 900 python tools/fit_throughput.py 2>&1 | tail -1 > gpurun_out/fit_throughput.json
timeout 600 python tools/perf_loss.py 2>&1 | tail -2 > gpurun_out/perf_loss.txt
