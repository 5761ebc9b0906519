cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 40 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
run base
CWF_DEFER_WGRAD=2 CWF_SIDE_WGS=128 run d2_s128
CWF_DEFER_WGRAD=2 CWF_SIDE_WGS=160 run d2_s160
CWF_DEFER_WGRAD=2 CWF_SIDE_WGS=96 run d2_s96
CWF_DEFER_WGRAD=2 CWF_SIDE_WGS=64 run d2_s64
CWF_DEFER_WGRAD=2 CWF_SIDE_WGS=128 CWF_FLUSH_EVERY=1000 run d2_s128_flushphase
CWF_DEFER_WGRAD=2 CWF_SIDE_WGS=128 run d2_s128_again
