set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "head_loss or upsample or losses" 2>&1 | tail -3
timeout -k 10 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "golden or trainer_training or grouped_head" 2>&1 | tail -3
python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"
