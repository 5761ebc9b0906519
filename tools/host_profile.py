"""cProfile of the host side of training steps (where do the ~25 ms of Python per step go?).  Diagnostic only."""
import cProfile, os, pstats, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "decouple-and-couple_learning_in_multi-modal_brain_tumor_segmentation_amd")); sys.path.insert(0, REPO)
import torch
from cwf import kernels
from cwf.trainer import Trainer
from models.clswiseformer.cls_wise_former import get_cls_wise_former
from utils import synthetic as syn
kernels.set_precision("bf16x3", wgrad="bf16", dgrad="bf16")
if os.environ.get("CWF_PROFILE_PG") == "1":          # with a one-rank RCCL process group alive (does it slow the host down?)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda:0")
m = get_cls_wise_former(dataset="brats", _conv_repr=True, _pe_type="fixed").to(dev).train()
tr = Trainer(m)
x, target, edge = syn.synthetic_batch([0, 1], (128, 128, 128))
x, target, edge = x.to(dev), target.to(dev), edge.to(dev)
for _ in range(3):
    tr.step(x, target, edge, 0)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(10):
    tr.step(x, target, edge, 0)
print("host enqueue ms/step (no profiler): %.2f" % ((time.perf_counter() - t0) * 100))
torch.cuda.synchronize()
if os.environ.get("CWF_PROFILE_BWD") == "1":
    torch.autograd.set_multithreading_enabled(False)      # backward on the calling thread: visible to cProfile
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    tr.step(x, target, edge, 0)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(40)
st.sort_stats("cumulative").print_stats(45)
