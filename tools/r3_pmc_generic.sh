set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "graph_replay or rccl_phase" > gpurun_out/r3_t3.log 2>&1 || true
tail -3 gpurun_out/r3_t3.log
bash tools/pmc_layer.sh r3pmc_32f 3 32 32 64 fwd 10
bash tools/pmc_layer.sh r3pmc_32d 3 32 32 64 dgrad 10
bash tools/pmc_layer.sh r3pmc_64f 3 64 64 32 fwd 10
bash tools/pmc_layer.sh r3pmc_128f 3 128 128 16 fwd 10
bash tools/pmc_layer.sh r3pmc_128d 3 128 128 16 dgrad 10
for t in 32f 32d 64f 128f 128d; do echo "== $t"; python tools/pmc_summary.py gpurun_out/r3pmc_$t conv_bf16_kernel; done > gpurun_out/r3_pmc1_summary.txt 2>&1
cat gpurun_out/r3_pmc1_summary.txt | head -80
