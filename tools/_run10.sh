set -e
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "head_loss" 2>&1 | tail -1
export CWF_BENCH_REHEARSE=1
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 4 --warmup 2 > gpurun_out/r2_rehearse.json 2> gpurun_out/r2_rehearse.err || { tail -30 gpurun_out/r2_rehearse.err; exit 1; }
tail -1 gpurun_out/r2_rehearse.json | cut -c1-400
unset CWF_BENCH_REHEARSE
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
