# Kernel trace of the plan-mode bench + per-stream timeline of one step (tools/timeline.py).   bash tools/r3_timeline.sh <tag>
set -e
TAG=${1:-r3tl}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace -f csv -d $R/gpurun_out/${TAG} -o t -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${TAG}.log 2>&1
cd $R
python tools/timeline.py $(ls gpurun_out/${TAG}/*kernel_trace.csv | head -1) > gpurun_out/${TAG}_timeline.txt
cat gpurun_out/${TAG}_timeline.txt
python bench.py --steps 30 --warmup 6 --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 600 gpurun_out/${TAG}_bench.json
