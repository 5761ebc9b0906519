#!/bin/bash
# Per-step launch census on the GPU box: two rocprofv3 kernel traces of bench.py that differ only in --steps (see launch_census.py).
#   bash tools/gpu_census.sh <tag> [extra bench args]        -> gpurun_out/<tag>_{a,b}/ , gpurun_out/<tag>_census.txt
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -f csv rocpd -d $ROOT/gpurun_out/${TAG}_a -o a -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" > $ROOT/gpurun_out/${TAG}_a.log 2>&1
rocprofv3 --kernel-trace --stats -f csv rocpd -d $ROOT/gpurun_out/${TAG}_b -o b -- python $ROOT/bench.py --steps 12 --warmup 2 --no-cpu-baseline "$@" > $ROOT/gpurun_out/${TAG}_b.log 2>&1
cd $ROOT
python tools/launch_census.py gpurun_out/${TAG}_a/a_kernel_stats.csv 4 gpurun_out/${TAG}_b/b_kernel_stats.csv 12 gpurun_out/${TAG}_census.csv > gpurun_out/${TAG}_census.txt
head -3 gpurun_out/${TAG}_census.txt
