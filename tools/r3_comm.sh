# The data-parallel path on the one-GPU box: a one-rank RCCL group with the per-phase all-reduces forced on (plan and eager mode) against no group.
cd $GRAFT_REPO_ROOT
run() { tag="$1"; shift; env "$@" python bench.py --steps 30 --warmup 6 --no-cpu-baseline $EXTRA 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s' % '$tag', j['value'], j['ms_per_step'], j.get('comm_forced'), j['config'].get('plan'))"; }
EXTRA=""
run nogroup A=1
run nogroup_q4 GPU_MAX_HW_QUEUES=4
run plan_forced RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 CWF_FORCE_COMM=1
run plan_group_only RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 CWF_FORCE_COMM=init
EXTRA="--mode eager"
run eager_nogroup A=1
run eager_forced RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29513 CWF_FORCE_COMM=1
