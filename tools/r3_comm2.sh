cd $GRAFT_REPO_ROOT
run() { tag="$1"; shift; env "$@" python bench.py --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s' % '$tag', j['value'], j['ms_per_step'], j.get('comm_forced'))"; }
C="RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 CWF_FORCE_COMM=1"
run plan_forced $C MASTER_PORT=29521
run plan_forced_q8 $C MASTER_PORT=29522 GPU_MAX_HW_QUEUES=8
run plan_forced_q2 $C MASTER_PORT=29523 GPU_MAX_HW_QUEUES=2
run nogroup_q8 GPU_MAX_HW_QUEUES=8
