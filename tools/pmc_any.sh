#!/bin/bash
# rocprofv3 --pmc over any tools/*.py micro-benchmark, one counter group per pass.
#   bash tools/pmc_any.sh <tag> <script.py> <args...>     -> gpurun_out/<tag>/g<i>/...  (summarise with tools/pmc_summary.py <dir> <kernel substring>)
# groups: FETCH_SIZE | WRITE_SIZE | MFMA busy / SQ busy / MFMA count / wave cycles | LDS + wait split | L1->L2 requests and L2 hits
set -e
TAG=$1; SCRIPT=$2; shift; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -f csv -d $R/gpurun_out/$TAG/g$i -o p -- python $R/tools/$SCRIPT "$@" > $R/gpurun_out/${TAG}_g$i.log 2>&1 || echo "group $i ($grp) failed" >> $R/gpurun_out/${TAG}_fail.log
done
