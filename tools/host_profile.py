"""cProfile of the host side of training steps (where do the ~25 ms of Python per step go?).  Diagnostic only."""
import cProfile, os, pstats, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "decouple-and-couple_learning_in_multi-modal_brain_tumor_segmentation_amd")); sys.path.insert(0, REPO)
import torch
from cwf import kernels
from cwf.trainer import Trainer
from models.clswiseformer.cls_wise_former import get_cls_wise_former
from utils import synthetic as syn
kernels.set_precision("bf16x3")
dev = torch.device("cuda:0")
m = get_cls_wise_former(dataset="brats", _conv_repr=True, _pe_type="fixed").to(dev).train()
tr = Trainer(m)
x, target, edge = syn.synthetic_batch([0, 1], (128, 128, 128))
x, target, edge = x.to(dev), target.to(dev), edge.to(dev)
for _ in range(3):
    tr.step(x, target, edge, 0)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    tr.step(x, target, edge, 0)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(22)
