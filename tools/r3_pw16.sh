set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "writes_its_bf16_image or pointwise" 2>&1 | tail -3
timeout -k 10 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -3
bash tools/r3_ab.sh A=1 CWF_NO_PW_Y16=1
