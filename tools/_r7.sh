set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace -f csv -d $R/gpurun_out/r2g -o s -- python $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline > $R/gpurun_out/r2g.log 2>&1
python $R/tools/timeline.py $R/gpurun_out/r2g/s_kernel_trace.csv
