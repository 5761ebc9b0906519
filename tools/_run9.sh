set -e
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "head_loss" 2>&1 | tail -2
timeout -k 10 600 python tools/conv_cfg_sweep.py bf16 bf16x3 > gpurun_out/r2_cfg_sweep.txt 2>&1
cat gpurun_out/r2_cfg_sweep.txt | grep " us "
