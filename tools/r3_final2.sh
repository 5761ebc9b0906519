set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/layer_table.py > gpurun_out/r3f_layer_table.txt 2>/dev/null
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/r3f_micro -o m -- python $R/tools/conv16_micro.py bf16x3 20 conv > $R/gpurun_out/r3f_micro.log 2>&1
cd $R
bash tools/r3_tl_stats.sh r3f_tl > /dev/null
python bench.py --steps 50 --warmup 10 > gpurun_out/r3f_bench.json 2> gpurun_out/r3f_bench.err
tail -c 400 gpurun_out/r3f_bench.json
