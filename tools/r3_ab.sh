# A/B of environment switches in ONE gpurun call (boxes differ by +-2 %): bash tools/r3_ab.sh "VAR=1" "VAR=0" ...  (each run twice, interleaved)
cd $GRAFT_REPO_ROOT
run() { tag="$1"; env $1 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s' % '$tag', j['value'], j['ms_per_step'])"; }
for rep in 1 2; do for v in "$@"; do run "$v"; done; done
