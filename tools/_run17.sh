timeout -k 10 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "sliding or flip" > gpurun_out/r2_t9.log 2>&1
tail -1 gpurun_out/r2_t9.log; grep -E "^E " gpurun_out/r2_t9.log | head
