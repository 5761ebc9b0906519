set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "weight_stationary or conv_family or fused_norm" > gpurun_out/r3_ws_t.log 2>&1 || true
tail -4 gpurun_out/r3_ws_t.log
python tools/ws_debug2.py bf16 2>&1 | grep -v amdgpu.ids
python tools/ws_debug2.py bf16x3 2>&1 | grep -v amdgpu.ids
for cfg in "3 32 32 64" "3 64 64 32"; do
  for mode in fwd dgrad; do
    python tools/layer_micro.py $cfg $mode 20 2>/dev/null
    CWF_NO_CONV_WS=1 python tools/layer_micro.py $cfg $mode 20 2>/dev/null | sed 's/^/   old: /'
  done
done 2>&1 | tee gpurun_out/r3_ws_micro.txt
