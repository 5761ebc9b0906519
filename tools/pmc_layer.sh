#!/bin/bash
# rocprofv3 --pmc over tools/layer_micro.py for one conv layer, one counter group per pass (FETCH_SIZE and WRITE_SIZE cannot share a
# pass on gfx950; SQ has 8 slots).   bash tools/pmc_layer.sh <tag> <layer_micro args...>   -> gpurun_out/<tag>/<group>/...
# The program sits directly after `--` (the profiler's preloaded library initialises the GPU before python starts).
set -e
TAG=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -f csv -d $R/gpurun_out/$TAG/g$i -o p -- python $R/tools/layer_micro.py "$@" > $R/gpurun_out/${TAG}_g$i.log 2>&1 || echo "group $i ($grp) failed" >> $R/gpurun_out/${TAG}_fail.log
done
