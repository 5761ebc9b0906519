set -e
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2_t6.log 2>&1 || { tail -40 gpurun_out/r2_t6.log | cut -c1-300; exit 1; }
tail -3 gpurun_out/r2_t6.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_c4_bench.json 2> gpurun_out/r2_c4_bench.err || { tail -20 gpurun_out/r2_c4_bench.err; exit 1; }
cat gpurun_out/r2_c4_bench.json | cut -c1-300
bash tools/gpu_census.sh r2_c4
