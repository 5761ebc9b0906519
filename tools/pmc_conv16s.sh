#!/bin/bash
# HBM traffic of the dominant kernel for bench.py's roofline.traffic: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE cannot share
# a pass on gfx950) over tools/conv16_micro.py; writes gpurun_out/<tag>/..., summarise with the snippet in profiles/*pmc.txt headers.
set -e
TAG=${1:-pmc_conv16s}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -f csv -d $R/gpurun_out/$TAG/conv_$c -o p -- python $R/tools/conv16_micro.py bf16x3 8 conv > $R/gpurun_out/${TAG}_$c.log 2>&1
done
