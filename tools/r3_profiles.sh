# Round-3 measurement batch for profiles/: per-layer table, per-step census, kernel stats of the bench and of the dominant-kernel micro-benchmark.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/layer_table.py > gpurun_out/r3_layer_table.txt 2>/dev/null
bash tools/gpu_census.sh r3fin
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/r3fin_micro -o m -- python $R/tools/conv16_micro.py bf16x3 20 conv > $R/gpurun_out/r3fin_micro.log 2>&1
cd $R
python bench.py --steps 50 --warmup 10 > gpurun_out/r3fin_bench.json 2> gpurun_out/r3fin_bench.err
tail -c 400 gpurun_out/r3fin_bench.json
