set -e
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "config4 or dataparallel or noncubic" > gpurun_out/r2_t7.log 2>&1 || { tail -40 gpurun_out/r2_t7.log | cut -c1-300; exit 1; }
tail -2 gpurun_out/r2_t7.log
for a in "" "--no-wgrad-async"; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
done
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -f csv rocpd -d $GRAFT_REPO_ROOT/gpurun_out/r2_serial -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-wgrad-async > $GRAFT_REPO_ROOT/gpurun_out/r2_serial.log 2>&1
