set -e
python tools/grad_err.py bf16x3:-:- bf16x3:-:- bf16x3:bf16:- bf16x3:bf16:bf16
for a in "" "--wgrad-precision bf16" "--wgrad-precision bf16 --dgrad-precision bf16"; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
done
