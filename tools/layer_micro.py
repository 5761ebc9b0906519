"""One conv layer's forward / data gradient in isolation, for rocprofv3 (kernel trace or --pmc).
usage: layer_micro.py OP(3|s|1|t) CIN COUT SIZE MODE(fwd|dgrad) [iters]      (batch 2, bench precisions).  Diagnostic."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "decouple-and-couple_learning_in_multi-modal_brain_tumor_segmentation_amd")); sys.path.insert(0, REPO)
import torch
from cwf import functional as CF, packing as pk, kernels
opk, cin, cout, s, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 20
kernels.set_precision("bf16x3", wgrad="bf16", dgrad="bf16")
K = kernels.backend()
dev = torch.device("cuda:0")
op = {"3": pk.CONV3_S1, "s": pk.CONV3_S2, "1": pk.CONV1, "t": pk.CONVT2}[opk]
n = 2
x = torch.randn((n, s, s, s, cin), device=dev)
wshape = (cin, cout, 2, 2, 2) if op == pk.CONVT2 else ((cout, cin, 1, 1, 1) if op == pk.CONV1 else (cout, cin, 3, 3, 3))
w = torch.nn.Parameter(torch.randn(wshape, device=dev) * 0.05)
spec = CF.ConvSpec(op, cin, cout)
packer = CF.WeightPacker(); packer.add(spec, w); packer.refresh()
b = torch.zeros(cout, device=dev)
y = K.conv(op, x, spec.packed(False), b, cout, w_ref=w)
dy = torch.randn_like(y); dx = torch.empty_like(x)
if mode == "fwd":
    run = lambda: K.conv(op, x, spec.packed(False), b, cout, out=y, w_ref=w)
else:
    run = lambda: K.conv(pk.dgrad_op(op), dy, spec.packed(True), None, cin, out=dx, w_ref=w, fwd_op=op)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): run()
e1.record(); torch.cuda.synchronize()
print("%s %s %d->%d s=%d: %.1f us" % (mode, opk, cin, cout, s, e0.elapsed_time(e1) / iters * 1e3))
