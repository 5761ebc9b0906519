# timeline + per-kernel mean durations of one traced plan-mode bench.   bash tools/r3_tl_stats.sh <tag> [env assignments...]
set -e
TAG=$1; shift
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/${TAG} -o t -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${TAG}.log 2>&1
cd $R
python tools/timeline.py $(ls gpurun_out/${TAG}/*kernel_trace.csv | head -1) > gpurun_out/${TAG}_timeline.txt
cat gpurun_out/${TAG}_timeline.txt
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/${TAG}/t_kernel_stats.csv')))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:28]:
    print('%6d %9.1f us  %8.1f us/launch  %s' % (int(r['Calls']), float(r['TotalDurationNs'])/1e3/9, float(r['AverageNs'])/1e3, r['Name'][:90]))
PY
