CWF_DIAG=1 timeout -k 10 200 python tools/conv16_micro.py bf16 10 conv 2>&1 | grep -E "mode|conv"
CWF_DIAG=1 timeout -k 10 200 python tools/conv16_micro.py bf16x3 10 conv 2>&1 | grep -E "mode|conv"
