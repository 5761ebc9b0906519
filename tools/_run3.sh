set -e
timeout -k 10 300 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "graph_replay or wgrad_side" > gpurun_out/r2_t4.log 2>&1 || { tail -40 gpurun_out/r2_t4.log; exit 1; }
tail -3 gpurun_out/r2_t4.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph > gpurun_out/r2_c2_graph.json 2> gpurun_out/r2_c2_graph.err || { tail -20 gpurun_out/r2_c2_graph.err; exit 1; }
cat gpurun_out/r2_c2_graph.json | cut -c1-200
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_c2_eager.json 2> gpurun_out/r2_c2_eager.err
cat gpurun_out/r2_c2_eager.json | cut -c1-200
