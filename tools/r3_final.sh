# Round-3 final measurement batch: GPU tests, PMC of the bf16-image kernels and the dominant kernel, per-step census, layer table, bench.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu > gpurun_out/r3f_gpu_all.log 2>&1 || true
tail -3 gpurun_out/r3f_gpu_all.log
bash tools/pmc_any.sh r3f_pmc_conv16s conv16_micro.py bf16x3 8 conv
bash tools/pmc_any.sh r3f_pmc_wgradd conv16_micro.py bf16 8 wgradd
bash tools/pmc_any.sh r3f_pmc_dgrad16 conv16_micro.py bf16 8 dgrad16
bash tools/pmc_any.sh r3f_pmc_apply16 conv16_micro.py bf16 8 apply16
for w in conv; do python tools/conv16_micro.py bf16x3 20 $w 2>/dev/null | tail -1; done
for w in wgradd dgrad16 dgrad16p apply16; do python tools/conv16_micro.py bf16 20 $w 2>/dev/null | tail -1; done
CWF_SIDE_WGS=256 python tools/conv16_micro.py bf16 20 wgradd 2>/dev/null | tail -1
python tools/pmc_summary.py gpurun_out/r3f_pmc_conv16s conv16s_kernel > gpurun_out/r3f_pmc_summary.txt
python tools/pmc_summary.py gpurun_out/r3f_pmc_wgradd wgrad16d_kernel >> gpurun_out/r3f_pmc_summary.txt
python tools/pmc_summary.py gpurun_out/r3f_pmc_dgrad16 conv16s_kernel >> gpurun_out/r3f_pmc_summary.txt
python tools/pmc_summary.py gpurun_out/r3f_pmc_apply16 in_bwd_apply_kernel >> gpurun_out/r3f_pmc_summary.txt
bash tools/gpu_census.sh r3f
python tools/layer_table.py > gpurun_out/r3f_layer_table.txt 2>/dev/null
bash tools/r3_tl_stats.sh r3f_tl > /dev/null
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/r3f_micro -o m -- python $R/tools/conv16_micro.py bf16x3 20 conv > $R/gpurun_out/r3f_micro.log 2>&1
cd $R
python bench.py --steps 50 --warmup 10 > gpurun_out/r3f_bench.json 2> gpurun_out/r3f_bench.err
tail -c 500 gpurun_out/r3f_bench.json
