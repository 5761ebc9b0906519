set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "weight_stationary or conv_family or fused_norm" > gpurun_out/r3_ws_t.log 2>&1 || true
tail -4 gpurun_out/r3_ws_t.log
for cfg in "3 32 32 64" "3 64 64 32" "3 128 128 16"; do
  for mode in fwd dgrad; do
    python tools/layer_micro.py $cfg $mode 20 2>/dev/null
  done
done 2>&1 | tee gpurun_out/r3_ws_micro.txt
for d in 1 2 4 8 3 6 7 15; do
  for mode in fwd dgrad; do
    CWF_WS_DIAG=$d python tools/layer_micro.py 3 32 32 64 $mode 20 2>/dev/null | sed "s/^/diag $d: /"
  done
done 2>&1 | tee -a gpurun_out/r3_ws_micro.txt
