set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -3
run() { tag=$1; shift; env "$@" python bench.py --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', j['value'], j['ms_per_step'], j['config'].get('plan'))"; }
run base A=1
run base2 A=1
