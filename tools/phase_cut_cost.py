"""Host time spent inside Trainer._phase_done (release of deferred weight gradients + batched reduce launch) per step.  Diagnostic."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "decouple-and-couple_learning_in_multi-modal_brain_tumor_segmentation_amd")); sys.path.insert(0, REPO)
import torch
from cwf import kernels
from cwf.trainer import Trainer
from models.clswiseformer.cls_wise_former import get_cls_wise_former
from utils import synthetic as syn
kernels.set_precision("bf16x3", wgrad="bf16", dgrad="bf16")
dev = torch.device("cuda:0")
m = get_cls_wise_former(dataset="brats", _conv_repr=True, _pe_type="fixed").to(dev).train()
tr = Trainer(m)
x, target, edge = [t.to(dev) for t in syn.synthetic_batch([0, 1], (128, 128, 128))]
K = kernels.backend()
acc = {}
def timed(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0; return r
    setattr(obj, name, g)
timed(K, "wgrad_release", "release"); timed(K, "wgrad_flush", "flush")
orig = tr._phase_done
def pd(k):
    t0 = time.perf_counter(); orig(k); acc["phase_done_%d" % k] = acc.get("phase_done_%d" % k, 0.0) + time.perf_counter() - t0
m.phase_callback = pd
for _ in range(3): tr.step(x, target, edge, 0)
torch.cuda.synchronize(); acc.clear()
n = 10
for _ in range(n): tr.step(x, target, edge, 0)
torch.cuda.synchronize()
for k, v in sorted(acc.items()): print("%-16s %.3f ms per step" % (k, v / n * 1e3))
