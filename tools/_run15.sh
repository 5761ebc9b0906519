set -e
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2_t8.log 2>&1 || { tail -40 gpurun_out/r2_t8.log | cut -c1-300; exit 1; }
tail -1 gpurun_out/r2_t8.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 400 python bench.py > gpurun_out/r2_final_bench.json 2> gpurun_out/r2_final_bench.err || { tail -20 gpurun_out/r2_final_bench.err; exit 1; }
cat gpurun_out/r2_final_bench.json | cut -c1-250
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('graph', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
bash tools/gpu_census.sh r2_final
