set -e
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r2_t5.log 2>&1 || { tail -40 gpurun_out/r2_t5.log | cut -c1-300; exit 1; }
tail -3 gpurun_out/r2_t5.log
timeout -k 10 200 python tools/find_fills.py > gpurun_out/r2_find_fills.txt 2>&1 || true
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2_c3_bench.json 2> gpurun_out/r2_c3_bench.err
cat gpurun_out/r2_c3_bench.json | cut -c1-300
bash tools/gpu_census.sh r2_c3
