"""Timings of the BASELINE.json configurations that are not the bench line (1 GPU): configs[3] sliding-window inference on a full
240x240x155 volume (predict_overlap.tailor_and_concat, 8 windows as one batch) and configs[4]'s 160x192x160 training patch (B = 1)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "decouple-and-couple_learning_in_multi-modal_brain_tumor_segmentation_amd")); sys.path.insert(0, REPO)
import torch
from cwf import kernels
from cwf.trainer import Trainer
from models.clswiseformer.cls_wise_former import get_cls_wise_former
from utils import synthetic as syn
import predict_overlap as po
dev = "cuda:0"
kernels.set_precision("bf16x3", wgrad="bf16", dgrad="bf16")
m = get_cls_wise_former(dataset="brats", _conv_repr=True, _pe_type="fixed").to(dev)
m.Unet_list.InitConv.dropout = 0.0
x = torch.randn(1, 4, 240, 240, 155, device=dev)
m.eval()
with torch.no_grad():
    for _ in range(2):
        y = po.tailor_and_concat(x, None, m)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        y = po.tailor_and_concat(x, None, m)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("configs[3] sliding-window inference 240x240x155 (8 windows of 128^3, one batch): %.1f ms per volume = %.2f volumes/s = %.1f window-forwards/s" % (dt * 1e3, 1 / dt, 8 / dt))
m.train(); m.Unet_list.InitConv.dropout = 0.2
tr = Trainer(m)
xb, tb, eb = [t.to(dev) for t in syn.synthetic_batch([0], (160, 192, 160))]
for _ in range(3):
    tr.step(xb, tb, eb, 0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    tr.step(xb, tb, eb, 0)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print("configs[4] training step on one 160x192x160 patch (2.34x the voxels of 128^3), all 12 sub-region/edge heads: %.1f ms per step = %.2f patches/s = %.1f 128^3-equivalents/s" % (dt * 1e3, 1 / dt, 2.34375 / dt))
print("peak memory %.1f GB" % (torch.cuda.max_memory_allocated() / 1e9))
