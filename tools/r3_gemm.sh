set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_coupler_gpu.py -x -q -m gpu -k "token or coupler or gemm or attention or layernorm or linear" 2>&1 | tail -3
echo NEW; python tools/gemm_census.py 2>/dev/null | tail -12
echo OLD; CWF_GEMM_V=1 python tools/gemm_census.py 2>/dev/null | tail -12
