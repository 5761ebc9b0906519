# bf16 operand images + DMA weight gradient: unit test, then the whole-model gradient tests, then the bench.
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "bf16_operand_images or instance_norm_pieces or fused_norm_backward" 2>&1 | tail -3
run() { tag=$1; shift; env "$@" python bench.py --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', j['value'], j['ms_per_step'], j['config'].get('plan'))"; }
run base A=1
run emits_xa_dx CWF_APPLY_EMITS=xa,dx
run off CWF_NO_BF16_OPERANDS=1
run wgs160 CWF_SIDE_WGS=160
run wgs192 CWF_SIDE_WGS=192
run defer1 CWF_DEFER_WGRAD=1
