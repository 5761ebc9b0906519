for d in 0 1; do
  CWF_DEFER_WGRAD=$d timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('defer=$d', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
done
