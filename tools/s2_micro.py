"""conv_s2.hip / conv_stem.hip in both operand forms (split-bf16 / single bf16): which chain bounds them.  Diagnostic."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "decouple-and-couple_learning_in_multi-modal_brain_tumor_segmentation_amd")); sys.path.insert(0, REPO)
import torch
from cwf import functional as CF, packing as pk, kernels
K = kernels.backend()
dev = torch.device("cuda:0")
def bench(op, cin, cout, prec):
    kernels.set_precision(prec)
    x = torch.randn((2, 128, 128, 128, cin), device=dev)
    w = torch.nn.Parameter(torch.randn((cout, cin, 3, 3, 3), device=dev) * 0.05)
    spec = CF.ConvSpec(op, cin, cout)
    packer = CF.WeightPacker(); packer.add(spec, w); packer.refresh()
    b = torch.zeros(cout, device=dev)
    st = K.new_stats(2, cout, dev)
    y = K.conv(op, x, spec.packed(False), b, cout, w_ref=w, stats=st)
    run = lambda: K.conv(op, x, spec.packed(False), b, cout, out=y, w_ref=w, stats=st)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print("op %d %d->%d %s: %.1f us" % (op, cin, cout, prec, e0.elapsed_time(e1) / 20 * 1e3))
for prec in ("bf16x3", "bf16"):
    bench(pk.CONV3_S2, 16, 32, prec)
    bench(pk.CONV3_S1, 4, 16, prec)
