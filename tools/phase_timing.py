"""Where does a training step spend its time?  Host enqueue time vs GPU time per phase (encoder / three region pipelines +
cross-region coupler / decoder / losses / backward / Adam), eager mode, bf16x3.  Diagnostic only."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "decouple-and-couple_learning_in_multi-modal_brain_tumor_segmentation_amd")); sys.path.insert(0, REPO)
import torch
from cwf import kernels
from cwf.trainer import Trainer, total_loss
from models.clswiseformer.cls_wise_former import get_cls_wise_former
from utils import synthetic as syn

kernels.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16x3")
dev = torch.device("cuda:0")
m = get_cls_wise_former(dataset="brats", _conv_repr=True, _pe_type="fixed").to(dev).train()
if len(sys.argv) > 2 and sys.argv[2] == "serial":
    m.parallel_regions = False
tr = Trainer(m, use_graph=False)
x, target, edge = syn.synthetic_batch([0, 1], (128, 128, 128))
x, target, edge = x.to(dev), target.to(dev), edge.to(dev)
for _ in range(3):
    tr.step(x, target, edge, 0)
torch.cuda.synchronize()

marks = []
def mark(name):
    ev = torch.cuda.Event(enable_timing=True); ev.record()
    marks.append((name, time.perf_counter(), ev))

bmarks = []
def bmark(name):
    def hook(grad):
        ev = torch.cuda.Event(enable_timing=True); ev.record()
        bmarks.append((name, time.perf_counter(), ev))
        return grad
    return hook

unet_fwd = m.Unet_list.forward
def unet_wrapped(*a, **k):
    out = unet_fwd(*a, **k); mark("encoder (U-Net down path)")
    if torch.is_grad_enabled():
        out[3].register_hook(bmark("bwd: region pipelines done (grad of x4 ready)"))
    return out
m.Unet_list.forward = unet_wrapped
enc = m.encode
def enc_wrapped(*a, **k):
    out = enc(*a, **k); mark("region pipelines + cross-region coupler")
    if torch.is_grad_enabled():
        out[3].register_hook(bmark("bwd: decoder + heads + losses done (grad of bottleneck ready)"))
    return out
m.encode = enc_wrapped

for rep in range(2):
    marks.clear(); bmarks.clear()
    torch.cuda.synchronize()
    mark("start")
    outs = m(x, None); mark("decoder + heads")
    loss, parts = total_loss(outs, target, edge); mark("losses")
    tr.opt.zero_grad()
    from cwf.kernels import backend as _b
    _b().wgrad_async = True
    loss.backward(); _b().wgrad_async = False; _b().join_wgrad_stream(); mark("backward")
    tr.opt.step(); mark("adam")
    torch.cuda.synchronize()
    t_end = time.perf_counter()
print("%-45s %10s %10s" % ("phase", "host ms", "GPU ms"))
for (n0, h0, e0), (n1, h1, e1) in zip(marks[:-1], marks[1:]):
    print("%-45s %10.2f %10.2f" % (n1, (h1 - h0) * 1e3, e0.elapsed_time(e1)))
print("%-45s %10.2f %10.2f" % ("total", (marks[-1][1] - marks[0][1]) * 1e3, marks[0][2].elapsed_time(marks[-1][2])))
loss_mark = [mk for mk in marks if mk[0] == "losses"][0]
bwd_mark = [mk for mk in marks if mk[0] == "backward"][0]
prev = loss_mark
for name, h, ev in bmarks + [("bwd: encoder done", bwd_mark[1], bwd_mark[2])]:
    print("  %-70s host +%6.2f ms   GPU +%6.2f ms" % (name, (h - prev[1]) * 1e3, prev[2].elapsed_time(ev)))
    prev = (name, h, ev)
print("wall incl. final sync: %.2f ms" % ((t_end - marks[0][1]) * 1e3))
