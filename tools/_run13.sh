set -e
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv_family" 2>&1 | tail -2
timeout -k 10 120 python tools/conv16_micro.py bf16 20 conv
timeout -k 10 120 python tools/conv16_micro.py bf16x3 20 conv
for cs in "32 64" "64 32"; do CWF_X=1 timeout -k 10 120 python tools/conv_cfg_sweep.py child $cs bf16; done
