"""Micro-benchmark of the dominant kernel (3x3x3 conv 16->16 @128^3 x2) for rocprofv3 PMC runs."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "decouple-and-couple_learning_in_multi-modal_brain_tumor_segmentation_amd")); sys.path.insert(0, REPO)
import torch
from cwf import functional as CF, packing as pk, kernels
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
what = sys.argv[3] if len(sys.argv) > 3 else "conv"
kernels.set_precision(prec)
K = kernels.backend()
dev = "cuda:0"
n, s, c = 2, 128, 16
x = torch.randn((n, s, s, s, c), device=dev)
w = torch.nn.Parameter(torch.randn((c, c, 3, 3, 3), device=dev) * 0.05)
b = torch.zeros(c, device=dev)
spec = CF.ConvSpec(pk.CONV3_S1, c, c)
packer = CF.WeightPacker(); packer.add(spec, w); packer.refresh()
sc = torch.ones((n, c), device=dev); sh = torch.zeros((n, c), device=dev)
y = torch.empty_like(x); dy = torch.randn_like(x)
stats = K.new_stats(n, c, dev)
# round 3: the bf16-image kernels (what: wgradd = wgrad16d_kernel, dgrad16 = conv16s IN16 with residual + norm-backward sums as in the
# step, dgrad16p = IN16 plain, apply16 = in_bwd_apply writing only the two bf16 images).  Run with prec "bf16".
if what in ("wgradd", "dgrad16", "dgrad16p", "apply16"):
    kernels.set_precision("bf16x3", wgrad="bf16", dgrad="bf16")
    packer.refresh()
    x16 = K.to_bf16(x, sc, sh, 0.0); dy16 = K.to_bf16(dy); res = torch.randn_like(x)
    dw = torch.zeros(w.numel(), device=dev); db = torch.zeros(c, device=dev)
    sums = K.new_stats(n, c, dev)
def run():
    if what == "conv":
        K.conv(pk.CONV3_S1, x, spec.packed(False), b, c, sc, sh, 0.0, None, None, stats, out=y)
    elif what == "wgradd":
        K._wgrad_to_impl("micro", pk.CONV3_S1, x, sc, sh, 0.0, dy, c, spec.inv_map, dw, db, None, x16, dy16)
        K._wg_pending.clear()
    elif what == "dgrad16":
        K.conv(pk.CONV3_S1, dy, spec.packed(True), None, c, out=y, fwd_op=pk.CONV3_S1, prec="bf16", x16=dy16, residual=res, stats=sums, nb=(x, sc, sh, 0.0))
    elif what == "dgrad16p":
        K.conv(pk.CONV3_S1, dy, spec.packed(True), None, c, out=y, fwd_op=pk.CONV3_S1, prec="bf16", x16=dy16)
    elif what == "apply16":
        K.in_bwd_apply16(dy, x, sc, sh, 0.0, sums, want_dx16=True, want_xa16=True, need_f32=False)
    else:
        K.wgrad(pk.CONV3_S1, x, sc, sh, 0.0, dy, c, spec.inv_map, True, w.numel())
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print("%s %s: %.4f ms  %.1f TFLOP/s alg  %.1f GB/s alg" % (what, prec, ms, 2.0 * 27 * c * c * n * s ** 3 / ms / 1e9, 2.0 * n * s ** 3 * c * 4 / ms / 1e6))
if os.environ.get("CWF_SMODE"):
    import ctypes
    K.lib.cwf_debug_conv16_mode.argtypes = [ctypes.c_int]; K.lib.cwf_debug_conv16_mode.restype = None
    for mode in (0, 2, 8, 10):
        K.lib.cwf_debug_conv16_mode(mode)
        run(); torch.cuda.synchronize()
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        print("sliding kernel, mode %d (2 = no loads, 8 = no epilogue): %.4f ms" % (mode, e0.elapsed_time(e1) / 5))
    K.lib.cwf_debug_conv16_mode(0)
if os.environ.get("CWF_DIAG"):
    import ctypes
    diag = torch.zeros((256, 8, 4), dtype=torch.int64, device=dev)
    K.lib.cwf_debug_conv16_diag.argtypes = [ctypes.c_void_p]; K.lib.cwf_debug_conv16_diag.restype = None
    K.lib.cwf_debug_conv16_diag(diag.data_ptr())
    K.lib.cwf_debug_conv16_mode.argtypes = [ctypes.c_int]; K.lib.cwf_debug_conv16_mode.restype = None
    for mode in (0, 1, 2, 4, 6, 8):
        K.lib.cwf_debug_conv16_mode(mode)
        run(); torch.cuda.synchronize()
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        print("diag build, mode %d: %.4f ms" % (mode, e0.elapsed_time(e1) / 5))
    tiles = 2 * 8192 / 256
    for mode in (0, 1, 2, 4, 6, 8):
        K.lib.cwf_debug_conv16_mode(mode)
        diag.zero_(); torch.cuda.synchronize()
        run(); torch.cuda.synchronize()
        d = diag.cpu().double()
        m = d[:, :4].mean((0, 1)) / tiles
        l = d[:, 4:].mean((0, 1)) / tiles
        print("mode %2d per tile ticks  MFMA waves: barrier %.0f  mfma %.0f  epilogue %.0f | loader waves: barrier %.0f  commit %.0f" % (mode, m[0], m[1], m[2], l[0], l[2]))
    K.lib.cwf_debug_conv16_mode(0)
    K.lib.cwf_debug_conv16_diag(None)
