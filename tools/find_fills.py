"""Who launches the small fill kernels?  Python-level call-site census of zeros / zero_ / fill_ / zeros_like / add / clone
during one training step (C++-internal fills of the autograd engine are the remainder vs the kernel count).  Diagnostic."""
import collections, os, sys, traceback
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
cnt = collections.Counter()
def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "brain_tumor_segmentation_amd" in fr.filename and "find_fills" not in fr.filename:
            return "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
    return "?"
def wrap(obj, name, tag):
    orig = getattr(obj, name)
    def f(*a, **k):
        cnt[(tag, site())] += 1
        return orig(*a, **k)
    setattr(obj, name, f)
wrap(torch, "zeros", "torch.zeros"); wrap(torch, "zeros_like", "torch.zeros_like"); wrap(torch, "ones", "torch.ones")
wrap(torch.Tensor, "zero_", "Tensor.zero_"); wrap(torch.Tensor, "fill_", "Tensor.fill_"); wrap(torch.Tensor, "clone", "Tensor.clone")
wrap(torch.Tensor, "contiguous", "Tensor.contiguous"); wrap(torch.Tensor, "__add__", "Tensor.__add__"); wrap(torch.Tensor, "__mul__", "Tensor.__mul__")
wrap(torch.Tensor, "copy_", "Tensor.copy_"); wrap(torch.Tensor, "to", "Tensor.to"); wrap(torch.Tensor, "float", "Tensor.float"); wrap(torch.Tensor, "sum", "Tensor.sum")
tr.step(x, target, edge, 0)
torch.cuda.synchronize()
for (tag, where), n in cnt.most_common(50):
    print("%4d  %-20s %s" % (n, tag, where))
