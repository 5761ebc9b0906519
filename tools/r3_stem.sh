set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "stem_kernel" 2>&1 | tail -4
timeout -k 10 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "golden or trainer_training" 2>&1 | tail -3
bash tools/r3_ab.sh A=1 CWF_NO_STEM_KERNEL=1
