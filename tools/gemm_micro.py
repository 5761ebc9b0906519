"""Plain strided-batched GEMM (cwf_gemm_ex without epilogue features) at the token path's shapes: isolates the core loop from dropout / GELU / rowsum work."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "decouple-and-couple_learning_in_multi-modal_brain_tumor_segmentation_amd")); sys.path.insert(0, REPO)
import torch
from cwf import kernels
K = kernels.backend()
dev = "cuda:0"
def bench(m, n, k, zb, trans_a=False):
    a = torch.randn((zb, k, m) if trans_a else (zb, m, k), device=dev)
    b = torch.randn((zb, k, n), device=dev)
    c = torch.empty((zb, m, n), device=dev)
    sa = (1, m, k * m, 0) if trans_a else (k, 1, m * k, 0)
    def run():
        K.gemm(a, sa, b, (n, 1, k * n, 0), c, (n, m * n, 0), m, n, k, zb=zb)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    ref = torch.bmm(a.transpose(1, 2) if trans_a else a, b)
    err = float((c - ref).abs().max() / ref.abs().max())
    print("M %5d N %5d K %5d ZB %d %s : %7.1f us  %6.1f TFLOP/s  err %.1e" % (m, n, k, zb, "T" if trans_a else "N", us, 2.0 * m * n * k * zb / us / 1e6, err))
for shp in ((516, 512, 512, 3, False), (512, 512, 516, 3, True), (516, 1536, 512, 3, False), (1536, 512, 516, 3, True), (258, 512, 512, 1, False), (516, 512, 1024, 3, False)):
    bench(*shp)
