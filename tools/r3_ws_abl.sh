cd $GRAFT_REPO_ROOT
export CWF_WS_X3=1
for mode in fwd dgrad; do python tools/layer_micro.py 3 32 32 64 $mode 20 2>/dev/null; done
for d in 1 2 4 8 3 6 7 15; do
  for mode in fwd dgrad; do
    CWF_WS_DIAG=$d python tools/layer_micro.py 3 32 32 64 $mode 20 2>/dev/null | sed "s/^/diag $d: /"
  done
done
