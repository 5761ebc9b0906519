cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" python bench.py --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', j['value'], j['ms_per_step'])"; }
run base A=1
run defer1 CWF_DEFER_WGRAD=1
run defer0 CWF_DEFER_WGRAD=0
run defer1_emit CWF_DEFER_WGRAD=1 CWF_APPLY_EMITS=xa,dx
run defer0_emit CWF_DEFER_WGRAD=0 CWF_APPLY_EMITS=xa,dx
run defer1_emit_160 CWF_DEFER_WGRAD=1 CWF_APPLY_EMITS=xa,dx CWF_SIDE_WGS=160
run defer1_emit_96 CWF_DEFER_WGRAD=1 CWF_APPLY_EMITS=xa,dx CWF_SIDE_WGS=96
run defer1_emitdx CWF_DEFER_WGRAD=1 CWF_APPLY_EMITS=dx
run base2 A=1
