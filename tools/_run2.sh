set -e
python -m pytest tests/test_coupler_gpu.py -x -q -m gpu > gpurun_out/r2_t2.log 2>&1 || { tail -40 gpurun_out/r2_t2.log; exit 1; }
tail -3 gpurun_out/r2_t2.log
python -m pytest tests -x -q -m gpu > gpurun_out/r2_t3.log 2>&1 || { tail -40 gpurun_out/r2_t3.log; exit 1; }
tail -3 gpurun_out/r2_t3.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_c1_bench.json 2> gpurun_out/r2_c1_bench.err
cat gpurun_out/r2_c1_bench.json
bash tools/gpu_census.sh r2_c1
