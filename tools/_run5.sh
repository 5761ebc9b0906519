set -e
for p in bf16x3 bf16; do
  CWF_SMODE=1 timeout -k 10 120 python tools/conv16_micro.py $p 20 conv
  timeout -k 10 120 python tools/conv16_micro.py $p 20 wgrad
done
