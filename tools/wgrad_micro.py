"""Time one generic weight-gradient shape (slab kernel + reduce) in isolation; run under rocprofv3 --kernel-trace --stats for the
per-kernel split.  usage: wgrad_micro.py C SIZE [prec] [iters] [op: 3|1|t] [COUT]   Diagnostic."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "decouple-and-couple_learning_in_multi-modal_brain_tumor_segmentation_amd")); sys.path.insert(0, REPO)
import torch
from cwf import functional as CF, packing as pk, kernels
c, s = int(sys.argv[1]), int(sys.argv[2])
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 30
opk = sys.argv[5] if len(sys.argv) > 5 else "3"
cout = int(sys.argv[6]) if len(sys.argv) > 6 else c
kernels.set_precision("bf16x3", wgrad=prec, dgrad="bf16")
K = kernels.backend()
dev = torch.device("cuda:0")
n = 2
op = {"3": pk.CONV3_S1, "1": pk.CONV1, "t": pk.CONVT2}[opk]
so = 2 * s if op == pk.CONVT2 else s
x = torch.randn((n, s, s, s, c), device=dev); dy = torch.randn((n, so, so, so, cout), device=dev)
wshape = (c, cout, 2, 2, 2) if op == pk.CONVT2 else ((cout, c, 1, 1, 1) if op == pk.CONV1 else (cout, c, 3, 3, 3))
w = torch.nn.Parameter(torch.randn(wshape, device=dev) * 0.05)
spec = CF.ConvSpec(op, c, cout)
packer = CF.WeightPacker(); packer.add(spec, w); packer.refresh()
sc = torch.ones((n, c), device=dev); sh = torch.zeros((n, c), device=dev)
run = lambda: K.wgrad(op, x, sc, sh, 0.01, dy, cout, spec.inv_map, spec.has_bias_map, w.numel())
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): run()
e1.record(); torch.cuda.synchronize()
print("wgrad %s op=%s %d->%d s=%d: %.1f us per call (slabs + reduce)" % (prec, opk, c, cout, s, e0.elapsed_time(e1) / iters * 1e3))
