"""conv_ws vs tap-table kernel on one layer: error per 4x4x16 tile.  Diagnostic."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "decouple-and-couple_learning_in_multi-modal_brain_tumor_segmentation_amd")); sys.path.insert(0, REPO)
import torch
from cwf import functional as CF, packing as pk, kernels
cin, cout = int(sys.argv[1]), int(sys.argv[2])
d, h, w_ = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
n = int(sys.argv[6]) if len(sys.argv) > 6 else 1
prec = sys.argv[7] if len(sys.argv) > 7 else "bf16"
kernels.set_precision("bf16x3" if prec == "bf16x3" else "bf16")
K = kernels.backend()
dev = "cuda:0"
torch.manual_seed(0)
x = torch.randn((n, d, h, w_, cin), device=dev)
w = torch.nn.Parameter(torch.randn((cout, cin, 3, 3, 3), device=dev) * 0.05)
spec = CF.ConvSpec(pk.CONV3_S1, cin, cout)
packer = CF.WeightPacker(); packer.add(spec, w); packer.refresh()
feat = sys.argv[8] if len(sys.argv) > 8 else ""
kw = {}
bias = None
if "p" in feat:
    kw.update(in_scale=torch.rand((n, cin), device=dev) + 0.5, in_shift=torch.randn((n, cin), device=dev), slope=0.01)
if "r" in feat:
    kw.update(residual=torch.randn((n, d, h, w_, cout), device=dev))
if "b" in feat:
    bias = torch.randn(cout, device=dev)
def call():
    k2 = dict(kw)
    if "s" in feat:
        k2["stats"] = K.new_stats(n, cout, dev)
    return K.conv(pk.CONV3_S1, x, spec.packed(False), bias, cout, prec=prec, **k2), k2.get("stats")
old = K.lib.cwf_debug_ws_min_units(1 << 30)
y_ref, st_ref = call()
y_ref = y_ref.clone()
K.lib.cwf_debug_ws_min_units(1)
for rep in range(2):
    y, st = call()
    if st is not None:
        torch.cuda.synchronize(); print("stats rel err", float(((st - st_ref).abs() / (st_ref.abs() + 1e-9)).max()))
    torch.cuda.synchronize()
    err = (y - y_ref).abs()
    print("rep", rep, "max err", float(err.max()), "ref max", float(y_ref.abs().max()), "nan", int(torch.isnan(y).sum()))
    e = err.nan_to_num(99.0).view(n, d // 4, 4, h // 4, 4, w_ // 16, 16, cout).amax(dim=(2, 4, 6, 7))
    bad = (e > 1e-2 * float(y_ref.abs().max())).nonzero().tolist()
    print("bad tiles (n, td, th, tw):", bad[:40], "of", e.numel())
    if bad:
        b0 = bad[0]
        sub = err.nan_to_num(99.0)[b0[0], b0[1] * 4:(b0[1] + 1) * 4, b0[2] * 4:(b0[2] + 1) * 4, b0[3] * 16:(b0[3] + 1) * 16]
        print("first bad tile: err by (plane, row):", sub.amax(dim=(2, 3)).tolist())
        print("err by w:", [round(v, 3) for v in sub.amax(dim=(0, 1, 3)).tolist()])
        print("err by channel:", [round(v, 3) for v in sub.amax(dim=(0, 1, 2)).tolist()])
