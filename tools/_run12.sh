set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-24)
  rocprofv3 --pmc $c -f csv -d $R/gpurun_out/r2_pmc/conv_$tag -o p -- python $R/tools/conv16_micro.py bf16x3 8 conv > $R/gpurun_out/r2_pmc_conv_$tag.log 2>&1
  rocprofv3 --pmc $c -f csv -d $R/gpurun_out/r2_pmc/dgrad_$tag -o p -- python $R/tools/conv16_micro.py bf16 8 conv > $R/gpurun_out/r2_pmc_dgrad_$tag.log 2>&1
  rocprofv3 --pmc $c -f csv -d $R/gpurun_out/r2_pmc/wgrad_$tag -o p -- python $R/tools/conv16_micro.py bf16 8 wgrad > $R/gpurun_out/r2_pmc_wgrad_$tag.log 2>&1
done
cd $R
find gpurun_out/r2_pmc -name "*counter_collection.csv" | head -3
