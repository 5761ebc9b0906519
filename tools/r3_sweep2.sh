cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" python bench.py --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', j['value'], j['ms_per_step'])"; }
run base A=1
run wgs96 CWF_SIDE_WGS=96
run wgs160 CWF_SIDE_WGS=160
run wgs192 CWF_SIDE_WGS=192
run defer1 CWF_DEFER_WGRAD=1
run defer0 CWF_DEFER_WGRAD=0
run off CWF_NO_BF16_OPERANDS=1
run base2 A=1
