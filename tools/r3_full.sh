set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/r3_gpu_all.log 2>&1 || true
tail -4 gpurun_out/r3_gpu_all.log
bash tools/pmc_any.sh r3pmc_conv16s conv16_micro.py bf16x3 8 conv
bash tools/pmc_any.sh r3pmc_conv16s_bf16 conv16_micro.py bf16 8 conv
bash tools/pmc_any.sh r3pmc_wgrad16 conv16_micro.py bf16 8 wgrad
bash tools/pmc_any.sh r3pmc_wgrads1 wgrad_micro.py 32 64 bf16 8
bash tools/pmc_any.sh r3pmc_ws32f layer_micro.py 3 32 32 64 fwd 10
bash tools/pmc_any.sh r3pmc_ws32d layer_micro.py 3 32 32 64 dgrad 10
python bench.py --steps 30 --warmup 6 --no-cpu-baseline > gpurun_out/r3_bench_plan.json 2> gpurun_out/r3_bench_plan.err
tail -c 300 gpurun_out/r3_bench_plan.json
